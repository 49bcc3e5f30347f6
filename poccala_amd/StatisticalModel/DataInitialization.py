"""Data holder base class: the part of StatisticalModel/DataInitialization.py the hot path uses
(add_data :92-95, clear_data :97-100, data/datasize :114-120).  The CSV loader is out of scope."""


class DataInitialization(object):
    def __init__(self):
        self.__data = []
        self.__data_size = 0

    def add_data(self, data):
        self.__data.extend(data)
        self.__data_size += len(data)

    def clear_data(self):
        self.__data = []
        self.__data_size = 0

    @property
    def data(self):
        return self.__data

    @property
    def datasize(self):
        return self.__data_size
