"""Mirror of the network-relevant part of M2/common.py:76-83 (MyConfig.set_network_info)."""


class MyConfig(object):
    def __init__(self):
        self.set_network_info()
        self.lr = 1e-3
        self.lr_step_size = 15
        self.sr = 14000

    def set_network_info(self):
        self.kernel_sizes = [(1, 7), (7, 1)] + [(5, 5)] * 12
        self.dilations = [(1, 1), (1, 1), (1, 1), (2, 1), (4, 1), (8, 1), (16, 1), (32, 1),
                          (1, 1), (2, 2), (4, 4), (8, 8), (16, 16), (32, 32)]
