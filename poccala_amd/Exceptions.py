"""The exceptions of the reference that the mirrored surface raises (Exceptions.py): DataDimensionError on the hot path
(StatisticalModel/Clustering.py:749-751) and UnitFileExistsError from AcousticModel.load_unit (AcousticModel.py:142-143), with the
reference's constructor signatures."""


class DataDimensionError(Exception):
    def __init__(self, dimension, data_dimension, log=None):
        self.dimension = dimension
        self.data_dimension = data_dimension
        self.log = log

    def __str__(self):
        info = 'data dimension %s does not match model dimension %s' % (self.data_dimension, self.dimension)
        if self.log is not None:
            self.log.note(info, cls='e')
        return info


class UnitFileExistsError(FileExistsError):
    """Exceptions.py:24-32."""

    def __init__(self, unit_type, log=None):
        self.unit_type = unit_type
        self.log = log

    def __str__(self):
        info = 'unit file %s does not exist' % self.unit_type
        if self.log is not None:
            self.log.note(info, cls='e')
        return info


class NullLog(object):
    """Any object with .note(msg, cls=, show_console=) is accepted as a logger (LogPrint.py:64)."""

    def note(self, content, cls='i', show_console=True):
        pass

    def close(self):
        pass
