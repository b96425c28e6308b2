"""Plain right-hand sides in the callback contract of the reference (f(y, t, yout, p) -> None, writes yout in
place), written by hand for the ingestion tests: the FORM is the one the reference's generator emits
(symbolic/sympy_tools.py:100-111,185-195 -- assignments from p[..] and y[..], then yout[i] = (expression)); the
models are this build's own (a two-step conversion chain with a Hill-type activation and a time-dependent input)."""
import math

import numpy as np

ordered_params = ['k_in', 'k_conv', 'K_half', 'k_out', 'tau']
n_vars = 3


def model(y, t, yout, p):
    k_in = p[0]
    k_conv = p[1]
    K_half = p[2]
    k_out = p[3]
    tau = p[4]

    _a = y[0]
    _b = y[1]
    _c = y[2]

    # rate laws as intermediates (the reference's model files have a "Rate Laws" section; a hand-written
    # function keeps them as local names)
    drive = k_in * (1.0 - math.exp(-t / tau))
    v_ab = k_conv * _a ** 2 / (K_half ** 2 + _a ** 2)
    v_bc = k_conv * _b

    yout[0] = (drive - v_ab)
    yout[1] = (v_ab - v_bc)
    yout[2] = (v_bc - k_out * np.sqrt(_c + 1.0) * _c)


def sens_model(y, t, yout, p):
    """Sensitivities w.r.t. k_in, k_conv, k_out (K_half and tau held fixed): layout n_vars + i*k + j."""
    k_in, k_conv, K_half, k_out, tau = p[0], p[1], p[2], p[3], p[4]
    model(y, t, yout, p)
    a, b, c = y[0], y[1], y[2]
    S = np.asarray(y[3:12]).reshape(3, 3)
    h = a ** 2 / (K_half ** 2 + a ** 2)
    dh = 2 * a * K_half ** 2 / (K_half ** 2 + a ** 2) ** 2
    Jy = np.array([[-k_conv * dh, 0.0, 0.0],
                   [k_conv * dh, -k_conv, 0.0],
                   [0.0, k_conv, -k_out * (np.sqrt(c + 1.0) + 0.5 * c / np.sqrt(c + 1.0))]])
    Jp = np.array([[1.0 - math.exp(-t / tau), -h, 0.0],
                   [0.0, h - b, 0.0],
                   [0.0, b, -np.sqrt(c + 1.0) * c]])
    yout[3:12] = (Jy @ S + Jp).ravel()
