"""Import alias: the product package lives in the directory
``differentialdynamicprogramming.jl_amd/`` (a name Python cannot import directly because of the
dot), so ``import ddp_amd`` exposes that directory as a regular package."""
import os as _os

_PKG = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))),
                     "differentialdynamicprogramming.jl_amd")
__path__ = [_PKG]
with open(_os.path.join(_PKG, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_PKG, "__init__.py"), "exec"))
