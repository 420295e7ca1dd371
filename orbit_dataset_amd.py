"""Import alias for the package directory `orbit-dataset_amd/` (a hyphen cannot appear in a Python module name).

`import orbit_dataset_amd` executes orbit-dataset_amd/__init__.py in this module and points `__path__` at
that directory, so `orbit_dataset_amd.model.few_shot_recognisers` etc. resolve to files under it.
"""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "orbit-dataset_amd")]
with open(_os.path.join(__path__[0], "__init__.py")) as _f:
    exec(compile(_f.read(), _f.name, "exec"))
del _f
