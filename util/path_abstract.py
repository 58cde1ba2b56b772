"""Contract of the path configuration object the entry scripts query (three static getters, as in the reference's
util/path_abstract.py).  The getters are generated from one table so that a subclass only has to fill in what it knows."""

GETTERS = ("db_root_dir", "save_root_dir", "models_dir")


def _unset(name):
    def getter():
        raise NotImplementedError("%s() is not configured" % name)
    getter.__name__ = name
    return staticmethod(getter)


PathAbstract = type("PathAbstract", (object,), dict({g: _unset(g) for g in GETTERS},
                                                    __doc__="Base class: every getter raises NotImplementedError until overridden."))
