class PathAbstract(object):
    """Interface of the path configuration (reference util/path_abstract.py)."""

    @staticmethod
    def db_root_dir():
        raise NotImplementedError

    @staticmethod
    def save_root_dir():
        raise NotImplementedError

    @staticmethod
    def models_dir():
        raise NotImplementedError
