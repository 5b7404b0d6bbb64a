"""Minimal ``Box`` so that the env surface works without gym installed (environment.py:18-27
uses gym.spaces.Box only as metadata).  If gym / gymnasium is importable its Box is used."""
import numpy as np

try:                                   # pragma: no cover - neither is installed in the image
    from gym.spaces import Box         # type: ignore
except Exception:                      # noqa: BLE001
    try:
        from gymnasium.spaces import Box   # type: ignore
    except Exception:                  # noqa: BLE001
        class Box:
            def __init__(self, low, high, shape=None, dtype=np.float64):
                self.dtype = np.dtype(dtype)
                if shape is None:
                    shape = np.shape(low)
                self.shape = tuple(shape)
                self.low = np.broadcast_to(np.asarray(low, self.dtype), self.shape).copy()
                self.high = np.broadcast_to(np.asarray(high, self.dtype), self.shape).copy()

            def contains(self, x):
                x = np.asarray(x)
                return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))

            def sample(self):
                return np.random.uniform(self.low, self.high).astype(self.dtype)

            def __repr__(self):
                return "Box(%s, %s, %s, %s)" % (self.low.min(), self.high.max(), self.shape, self.dtype)
