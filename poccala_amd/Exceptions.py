"""The one exception of the reference that is raised on the hot path (Exceptions.py, raised at
StatisticalModel/Clustering.py:749-751), with the reference's constructor signature."""


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


class NullLog(object):
    """Any object with .note(msg, cls=, show_console=) is accepted as a logger (LogPrint.py:64)."""

    def note(self, content, cls='i', show_console=True):
        pass

    def close(self):
        pass
