"""Print the parity metrics of the HIP path vs the CPU oracle (run on the GPU box)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from oracle import step_ref
which = sys.argv[1] if len(sys.argv) > 1 else "all"
if which in ("tiny", "all"):
    print("tiny one-iter:", json.dumps(step_ref.compare_one_iteration(vqgan_model="tiny_f4", clip_model="tiny-B/32", size=(64, 64), cutn=8, seed=0)), flush=True)
if which in ("full", "all"):
    print("full one-iter:", json.dumps(step_ref.compare_one_iteration()), flush=True)
if which in ("steps", "all"):
    print("full 10 steps:", json.dumps(step_ref.compare_k_steps(10)), flush=True)
