"""Where the data, the snapshots and the pretrained weights live.  Same object protocol as the reference's mypath.py
(``Path.db_root_dir()``, ``Path.save_root_dir()``, ``Path.models_dir()``); each location can be overridden through an
environment variable so the entry scripts can be pointed at data without editing this file."""
import os

from util.path_abstract import GETTERS, PathAbstract

_LOCATIONS = {
    "db_root_dir": ("OSVOS_DB_ROOT", "/path/to/DAVIS-2016"),
    "save_root_dir": ("OSVOS_SAVE_ROOT", "./models"),
    "models_dir": ("OSVOS_MODELS_DIR", "./models"),
}
assert set(_LOCATIONS) == set(GETTERS)


def _lookup(env, default):
    return staticmethod(lambda: os.environ.get(env, default))


Path = type("Path", (PathAbstract,), {name: _lookup(*spec) for name, spec in _LOCATIONS.items()})
