"""`python -m pixray_amd --prompts "..." --quality draft ...`: the reference's command line (pixray.py:2126-2135) on this package."""
from .frontend import main

main()
