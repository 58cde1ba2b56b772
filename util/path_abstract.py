"""Contract of the path configuration object the entry scripts query: three static getters (same protocol as the
reference's util/path_abstract.py)."""


class PathAbstract(object):
    """Every getter raises NotImplementedError until a subclass (mypath.Path) overrides it."""

    @staticmethod
    def db_root_dir():
        raise NotImplementedError("db_root_dir() is not configured")

    @staticmethod
    def save_root_dir():
        raise NotImplementedError("save_root_dir() is not configured")

    @staticmethod
    def models_dir():
        raise NotImplementedError("models_dir() is not configured")
