"""Registration of custom-loss plugins, as the reference's `do_init` does it (pixray.py:131-140, 961-995, 2104-2109).

`--custom_loss "style:0.5,saturation->arg1->arg2"` is a comma-separated list of `name[:weight[:stop]]` chunks, optionally
followed by `->`-separated instance arguments.  Each name is looked up in `loss_class_table`, instantiated with
`device=`, given its instance arguments, then every instance parses the settings (`parse_settings(args) -> args`) and
contributes its globals (`add_globals(args) -> dict`).  The result is the list of {"loss", "weight"} dicts `Session` takes
as `custom_losses=` and the `lossGlobals` dict handed to every `get_loss` call.

Only `StyleLoss` ships with this package (the other reference losses are plain torch code and drop in unchanged through
`add_custom_loss`; see tests/test_host_logic.py::test_unmodified_reference_plugins_drop_in)."""
from typing import Dict, List, Tuple

from .interfaces import LossInterface
from .prompt import parse_prompt
from .style_loss import StyleLoss

loss_class_table: Dict[str, type] = {"style": StyleLoss}


def add_custom_loss(name: str, customloss: type) -> None:
    """pixray.py:2104-2109"""
    assert issubclass(customloss, LossInterface)
    loss_class_table.update({name: customloss})


def setup_custom_losses(spec, args, device=None) -> Tuple[List[dict], dict, object]:
    """pixray.py:961-995 -> (custom_losses, lossGlobals, args).  `spec` is the --custom_loss string (or None); an unknown
    name is a KeyError, a constructor that does not take `device=` re-raises its TypeError after the reference's hint."""
    losses: List[dict] = []
    loss_globals: dict = {}
    if spec:
        for chunk in [c.strip() for c in spec.split(",")]:
            if chunk.find("->") > 0:
                parts = chunk.split("->")
                name_part, instance_args = parts[0], parts[1:]
            else:
                name_part, instance_args = chunk, []
            loss_name, weight, _stop = parse_prompt(name_part)
            cls = loss_class_table[loss_name]
            try:
                inst = cls(device=device)
                inst.instance_settings(instance_args)
                losses.append({"loss": inst, "weight": weight})
            except TypeError as e:
                print(f"error in initializing {cls} - this message is to provide information")
                raise TypeError(e)
    for t in losses:
        args = t["loss"].parse_settings(args)
    for t in losses:
        loss_globals.update(t["loss"].add_globals(args))
    return losses, loss_globals, args
