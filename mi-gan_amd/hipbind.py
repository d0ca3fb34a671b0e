# placeholder, filled in below
class MiganError(RuntimeError):
    pass
class MiganLib:  # noqa
    pass
def load_library(path=None):
    raise MiganError("not built")
def library_path():
    return None
