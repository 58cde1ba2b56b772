"""Path configuration, same three getters as the reference's mypath.py.  Environment variables
override the defaults so the entry scripts can be pointed at data without editing files."""
import os

from util.path_abstract import PathAbstract


class Path(PathAbstract):
    @staticmethod
    def db_root_dir():
        return os.environ.get('OSVOS_DB_ROOT', '/path/to/DAVIS-2016')

    @staticmethod
    def save_root_dir():
        return os.environ.get('OSVOS_SAVE_ROOT', './models')

    @staticmethod
    def models_dir():
        return os.environ.get('OSVOS_MODELS_DIR', './models')
