"""Registration of drawer / filter / custom-loss plugins, as the reference's module tables and `do_init` do it
(pixray.py:54-58, 72-99, 131-140, 612-626, 650-669, 961-995, 2104-2109).

`--custom_loss "style:0.5,saturation->arg1->arg2"` is a comma-separated list of `name[:weight[:stop]]` chunks, optionally
followed by `->`-separated instance arguments.  Each name is looked up in `loss_class_table`, instantiated with
`device=`, given its instance arguments, then every instance parses the settings (`parse_settings(args) -> args`) and
contributes its globals (`add_globals(args) -> dict`).  The result is the list of {"loss", "weight"} dicts `Session` takes
as `custom_losses=` and the `lossGlobals` dict handed to every `get_loss` call.

`class_table` maps `--drawer` names to drawer classes (the HIP VQGAN drawer, the fft spectrum drawer, and the nearest-
upsampled pixel grid that stands in for the reference's `fast_pixel`); `filters_class_table` starts empty (the reference's
filters are plain torch modules and register themselves through `add_custom_filter`).

Only `StyleLoss` ships with this package (the other reference losses are plain torch code and drop in unchanged through
`add_custom_loss`; see tests/test_host_logic.py::test_unmodified_reference_plugins_drop_in)."""
from typing import Dict, List, Tuple

from .fft_drawer import FftDrawer
from .interfaces import DrawingInterface, FilterInterface, LossInterface
from .pixel_grid_drawer import PixelGridDrawer
from .prompt import parse_prompt
from .style_loss import StyleLoss
from .vqgan_drawer import VqganDrawer

class_table: Dict[str, type] = {"vqgan": VqganDrawer, "fft": FftDrawer, "fast_pixel": PixelGridDrawer}
filters_class_table: Dict[str, type] = {}
loss_class_table: Dict[str, type] = {"style": StyleLoss}


def add_custom_drawer(name: str, customdrawer: type) -> None:
    assert issubclass(customdrawer, DrawingInterface)
    class_table.update({name: customdrawer})


def add_custom_filter(name: str, customfilter: type) -> None:
    assert issubclass(customfilter, FilterInterface)
    filters_class_table.update({name: customfilter})


def make_drawer(args, device):
    """pixray.py:612-626 -> (drawer, (sideX, sideY)): the class named by `args.drawer` (KeyError when unknown), its model
    loaded, and the canvas size rounded down to a multiple of 2^(num_resolutions-1) when the drawer has resolutions"""
    drawer = class_table[args.drawer](args)
    drawer.load_model(args, device)
    num_resolutions = drawer.get_num_resolutions() if hasattr(drawer, "get_num_resolutions") else None
    if num_resolutions is not None:
        f = 2 ** (num_resolutions - 1)
        side = ((args.size[0] // f) * f, (args.size[1] // f) * f)
    else:
        side = (args.size[0], args.size[1])
    return drawer, side


def setup_filters(spec, args, device=None) -> List[dict]:
    """pixray.py:650-669: `--filters "name[:weight],..."` -> [{"filter", "weight"}]; an unknown name is a ValueError"""
    out: List[dict] = []
    if spec:
        for filt in [f.strip() for f in spec.split(",")]:
            filt_name, weight, _stop = parse_prompt(filt)
            if filt_name not in filters_class_table:
                raise ValueError(f"Requested filter not found, aborting: {filt_name}")
            cls = filters_class_table[filt_name]
            try:
                out.append({"filter": cls(args, device=device), "weight": weight})
            except TypeError as e:
                print(f"error in initializing {cls} - this message is to provide information")
                raise TypeError(e)
    return out


def add_custom_loss(name: str, customloss: type, needs_full_batch: bool = None) -> None:
    """pixray.py:2104-2109.  `needs_full_batch`: declare the class batch-coupled (engine.needs_full_batch) -- a loss written
    for this package sets the class attribute itself; the reference's own batch-coupled plugins (SaturationLoss,
    AestheticLoss, ResmemLoss) cannot, so they are marked HERE, at registration, by their reference class name."""
    assert issubclass(customloss, LossInterface)
    from .engine import REFERENCE_FULL_BATCH_LOSSES, mark_full_batch
    if needs_full_batch or (needs_full_batch is None and "needs_full_batch" not in vars(customloss)
                            and customloss.__name__ in REFERENCE_FULL_BATCH_LOSSES):
        mark_full_batch(customloss)
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
