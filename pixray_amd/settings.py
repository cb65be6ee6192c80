"""Host-side setting helpers the loop needs, with the behaviour the reference's own tests pin
(/root/reference/tests/test_util.py, tests/test_pixray.py): `parse_unit` (util.py:49-65), `split_pipes`
(util.py:67-71), `get_file_path` (util.py:32-37), `apply_overlay` (pixray.py:1431-1434),
`get_learning_rate_drops` (pixray.py:1999-2003).  The full argparse/yaml front end (pixray.py:1718-2116) is
outside the hot-path scope."""
import re
from pathlib import Path


def parse_unit(value, total_iterations, argument_name, default_unit="%"):
    """'50', '50%', '20 percent' -> fraction of total_iterations; '30i', '30 iterations' -> absolute."""
    if value is None:
        return None
    text = str(value).lower().strip()
    number = re.search(r"^\d*[.]?\d+", text)
    if re.match(r"^\d*[.]?\d+$", text):
        text += default_unit
    if re.match(r"^\d*[.]?\d+[\s]*(i|iter|iterations)$", text):
        return int(float(number.group(0)))
    if re.match(r"^\d*[.]?\d+[\s]*(p|%|percent)$", text):
        return int(float(number.group(0)) * 0.01 * total_iterations)
    raise ValueError(f"Invalid value for {argument_name}, please use a digit-unit combination like "
                     f"'20 iterations' or '50%'.")


def split_pipes(attribute):
    if not attribute:
        return attribute
    return [phrase.strip() for phrase in attribute.split("|")]


def get_file_path(directory, filename, suffix):
    """<directory>/<filename> with its extension replaced by `suffix`; bare names only (no separators)."""
    name = "" if filename is None else filename.strip()
    if name == "" or "/" in name or "\\" in name:
        raise ValueError("Invalid filename specified.")
    return str(Path(directory, filename).with_suffix(suffix))


def apply_overlay(args, cur_it):
    return args.overlay_image is not None and \
        (cur_it % args.overlay_every) == args.overlay_offset and \
        ((args.overlay_until is None) or (cur_it < args.overlay_until))


def get_learning_rate_drops(learning_rate_drops, iterations):
    if learning_rate_drops is None:
        return []
    return [parse_unit(n, iterations - 1, "learning_rate_drops") for n in learning_rate_drops]
