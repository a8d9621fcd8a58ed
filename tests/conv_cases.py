"""Literal known answers of the reference's array-convolution tests (test/dsp.jl), shared by the oracle test (CPU) and the
device parity test (GPU)."""
import numpy as np

# test/dsp.jl:130-146 ("conv-2D")
A2 = np.array([[1, 2, 1], [2, 3, 1], [1, 2, 1]])
B2 = np.array([[3, 2], [0, 1]])
EXP2 = np.array([[3, 8, 7, 2], [6, 14, 11, 3], [3, 10, 10, 3], [0, 1, 2, 1]])
IM_EXP2 = np.array([[3, 5, 5, 2], [3, 6, 6, 3], [3, 6, 6, 3], [0, 1, 1, 1]])     # imag of conv(complex.(a, 1), complex.(b))

# test/dsp.jl:203-226 ("separable conv")
SEP_U = np.array([1, 2, 3, 2, 1])
SEP_V = np.array([6, 7, 3, 2])
SEP_A = np.arange(1, 29).reshape(4, 7)
SEP_EXP = np.array([[6, 19, 35, 53, 71, 89, 107, 77, 33, 14],
                    [60, 148, 217, 285, 339, 393, 447, 315, 134, 56],
                    [204, 478, 658, 822, 930, 1038, 1146, 798, 338, 140],
                    [468, 1062, 1400, 1684, 1828, 1972, 2116, 1456, 614, 252],
                    [636, 1426, 1848, 2188, 2332, 2476, 2620, 1792, 754, 308],
                    [624, 1388, 1778, 2082, 2190, 2298, 2406, 1638, 688, 280],
                    [354, 785, 1001, 1167, 1221, 1275, 1329, 903, 379, 154],
                    [132, 292, 371, 431, 449, 467, 485, 329, 138, 56]])

# test/dsp.jl:229-253 ("conv-ND"): reshape(1:27, (3,3,3)) with ones(2,2,2); Julia reshapes column-major
A3 = np.arange(1, 28).reshape((3, 3, 3), order="F")
B3 = np.ones((2, 2, 2), dtype=np.int64)
EXP3 = np.array([1, 3, 5, 3, 5, 12, 16, 9, 11, 24, 28, 15, 7, 15, 17, 9,
                 11, 24, 28, 15, 28, 60, 68, 36, 40, 84, 92, 48, 23, 48, 52, 27,
                 29, 60, 64, 33, 64, 132, 140, 72, 76, 156, 164, 84, 41, 84, 88, 45,
                 19, 39, 41, 21, 41, 84, 88, 45, 47, 96, 100, 51, 25, 51, 53, 27]).reshape((4, 4, 4), order="F")


def promoted_case():
    """test/dsp.jl:261-268: a 3-d array against a matrix (trailing singleton promotion)."""
    a = np.stack([np.full((3, 3), n) for n in range(1, 7)], axis=2)
    b = np.ones((2, 2), dtype=np.int64)
    return a, b


# test/dsp.jl:317-360 ("xcorr"): (u, v, keyword arguments, expected)
XCORR = [
    ([1, 2], [3, 4], {}, [4, 11, 6]),
    ([1, 2, 3], [4, 5], {}, [5, 14, 23, 12]),
    ([1, 2, 3], [4, 5], {"padmode": "longest"}, [0, 5, 14, 23, 12]),
    ([1, 2, 3], [4, 5], {"padmode": "none"}, [5, 14, 23, 12]),
    ([1, 2], [3, 4, 5], {}, [5, 14, 11, 6]),
    ([1, 2], [3, 4, 5], {"padmode": "longest"}, [5, 14, 11, 6, 0]),
    ([1.0j], [1.0j], {}, [1]),
    (np.array([1, 2, 3]) * 1.0j, np.array([4, 5], dtype=np.complex128), {}, np.array([5, 14, 23, 12]) * 1j),
    (np.array([1, 2]) * 1.0j, np.array([3, 4, 5], dtype=np.complex128), {}, np.array([5, 14, 11, 6]) * 1j),
    (np.array([1, 2, 3], dtype=np.complex128), np.array([4, 5]) * 1.0j, {}, -np.array([5, 14, 23, 12]) * 1j),
    (np.array([1, 2, 3]) * 1.0j, np.array([4, 5]) * 1.0j, {}, [5, 14, 23, 12]),
    ([1, 2], [3, 4], {"scaling": "biased"}, [2.0, 5.5, 3.0]),
]
