"""Where the data, the snapshots and the pretrained weights live.  Same object protocol as the reference's mypath.py
(``Path.db_root_dir()``, ``Path.save_root_dir()``, ``Path.models_dir()``); each location can be overridden through an
environment variable so the entry scripts can be pointed at data without editing this file."""
import os

from util.path_abstract import PathAbstract


class Path(PathAbstract):
    @staticmethod
    def db_root_dir():
        return os.environ.get("OSVOS_DB_ROOT", "/path/to/DAVIS-2016")

    @staticmethod
    def save_root_dir():
        return os.environ.get("OSVOS_SAVE_ROOT", "./models")

    @staticmethod
    def models_dir():
        return os.environ.get("OSVOS_MODELS_DIR", "./models")
