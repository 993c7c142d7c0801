class SRLError(Exception):
    pass


class UndefinedError(SRLError):
    pass


class NotSupportedError(SRLError):
    pass
