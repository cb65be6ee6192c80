"""The reference's plugin base classes, kept source-compatible so existing drawers / custom losses /
filters drop in unchanged:

  DrawingInterface   /root/reference/DrawingInterface.py:1-9   (the de-facto API is duck-typed, see
                     vqgan.py:83-214 and the call sites listed in VqganDrawer below)
  LossInterface      /root/reference/Losses/LossInterface.py:4-35
  FilterInterface    /root/reference/filters/FilterInterface.py:4-16

`install_compat_modules()` registers these classes under the reference's module names
(`DrawingInterface`, `Losses.LossInterface`, `filters.FilterInterface`) so an unmodified plugin file
doing `from Losses.LossInterface import LossInterface` imports against this package.
"""
import argparse
import sys
import types

from torch import nn


class DrawingInterface:
    @staticmethod
    def add_settings(parser):
        return parser

    model = None

    def load_model(self, config, checkpoint):
        pass


class LossInterface:
    def __init__(self, device=None):
        self.device = device

    def instance_settings(self, arglist):
        pass

    @staticmethod
    def add_settings(parser):
        return parser

    def help(self):
        parser = argparse.ArgumentParser()
        parser = self.add_settings(parser)
        helpstring = ""
        for d in parser._actions:
            helpstring = f"""parmeter name: {d.dest}\nHelp: {d.help}\nUse case: pixray.add_argument({d.dest}={d.default})"""
        return helpstring

    def parse_settings(self, args):
        return args

    def add_globals(self, args):
        return {}

    def get_loss(self, cur_cutouts, out, args, globals=None, lossGlobals=None):
        return None


class FilterInterface(nn.Module):
    @staticmethod
    def add_settings(parser):
        return parser

    def __init__(self, settings, device=None):
        super().__init__()
        self.device = device

    def forward(self, img):
        return img, 0


def install_compat_modules():
    """Expose the interfaces under the reference's import paths (idempotent)."""
    m = types.ModuleType("DrawingInterface")
    m.DrawingInterface = DrawingInterface
    sys.modules.setdefault("DrawingInterface", m)
    for pkg, mod, cls in (("Losses", "LossInterface", LossInterface), ("filters", "FilterInterface", FilterInterface)):
        p = sys.modules.get(pkg)
        if p is None:
            p = types.ModuleType(pkg)
            p.__path__ = []
            sys.modules[pkg] = p
        sub = types.ModuleType(f"{pkg}.{mod}")
        setattr(sub, mod, cls)
        sys.modules.setdefault(f"{pkg}.{mod}", sub)
        setattr(p, mod, sys.modules[f"{pkg}.{mod}"])
