"""Small helpers of gaussfiltax/utils.py that sit on the filtering path's edges."""
import numpy as np


def mse(x_est, x_base):
    """gaussfiltax/utils.py:179-182."""
    x_est, x_base = np.asarray(x_est), np.asarray(x_base)
    return np.sum((x_est - x_base) ** 2) / x_est.shape[0]


def rmse(x_est, x_base):
    """gaussfiltax/utils.py:184-187."""
    x_est, x_base = np.asarray(x_est), np.asarray(x_base)
    return np.sqrt(np.sum((x_est - x_base) ** 2) / x_est.shape[0])
