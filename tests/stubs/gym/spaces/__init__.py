import numpy as np


class Space:
    def __init__(self, shape=None, dtype=None):
        self.shape = None if shape is None else tuple(shape)
        self.dtype = None if dtype is None else np.dtype(dtype)


class Box(Space):
    def __init__(self, low, high, shape=None, dtype=np.float32):
        if shape is None:
            low, high = np.asarray(low), np.asarray(high)
            assert low.shape == high.shape
            shape = low.shape
        super().__init__(shape, dtype)
        self.low = np.broadcast_to(np.asarray(low), self.shape).astype(self.dtype)
        self.high = np.broadcast_to(np.asarray(high), self.shape).astype(self.dtype)

    def contains(self, x):
        x = np.asarray(x)
        return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))

    def __repr__(self):
        return "Box" + str(self.shape)
