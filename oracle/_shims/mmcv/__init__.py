"""Import shim (test tooling only): lets /root/reference import without mmcv."""


class Config(dict):
    @staticmethod
    def fromfile(path):
        raise NotImplementedError("mmcv.Config shim: config files are restated by hand")
