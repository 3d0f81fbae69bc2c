"""Weight schema of the VNect graph and a seeded synthetic generator.

The reference stores weights as a pickle dict ``name -> ndarray`` produced by
``src/caffe2pkl.py:51-80`` and consumed by ``src/vnect_model.py:219-236``:

* ``'<scope>/weights'`` (kh, kw, Cin, Cout) + ``'<scope>/biases'`` (Cout) for every
  ``tc.layers.conv2d`` scope (``vnect_model.py:27-185, 211``),
* ``'<scope>/kernel'`` for the three ``tf.layers`` ops without bias
  (``vnect_model.py:188-193, 213``); transposed convs are (kh, kw, Cout, Cin),
* ``'bn5c_branch2a/{gamma,beta,moving_mean,moving_variance}'`` (``vnect_model.py:194``).

No trained weights ship with the reference (``models/*/README.md``), so parity is
established on seeded synthetic weights generated here.  The generator is a
counter-based integer hash (splitmix64) so that any implementation can regenerate
the same bits without numpy's RNG.
"""
import pickle

import numpy as np

MASTER_SEED = 20240807

# (scope, kernel, Cin, Cout, activation) for every tc.layers.conv2d scope, in the
# order of src/vnect_model.py.  activation: 'relu' (contrib default) or 'none'.
CONV_LAYERS = [
    ("conv1", 7, 3, 64, "relu"),
    ("res2a_branch1", 1, 64, 256, "none"),
    ("res2a_branch2a", 1, 64, 64, "relu"),
    ("res2a_branch2b", 3, 64, 64, "relu"),
    ("res2a_branch2c", 1, 64, 256, "none"),
    ("res2b_branch2a", 1, 256, 64, "relu"),
    ("res2b_branch2b", 3, 64, 64, "relu"),
    ("res2b_branch2c", 1, 64, 256, "none"),
    ("res2c_branch2a", 1, 256, 64, "relu"),   # dead in the reference graph (vnect_model.py:54-56)
    ("res2c_branch2b", 3, 64, 64, "relu"),
    ("res2c_branch2c", 1, 64, 256, "none"),
    ("res3a_branch1", 1, 256, 512, "none"),
    ("res3a_branch2a", 1, 256, 128, "relu"),
    ("res3a_branch2b", 3, 128, 128, "relu"),
    ("res3a_branch2c", 1, 128, 512, "none"),
    ("res3b_branch2a", 1, 512, 128, "relu"),
    ("res3b_branch2b", 3, 128, 128, "relu"),
    ("res3b_branch2c", 1, 128, 512, "none"),
    ("res3c_branch2a", 1, 512, 128, "relu"),
    ("res3c_branch2b", 3, 128, 128, "relu"),
    ("res3c_branch2c", 1, 128, 512, "none"),
    ("res3d_branch2a", 1, 512, 128, "relu"),
    ("res3d_branch2b", 3, 128, 128, "relu"),
    ("res3d_branch2c", 1, 128, 512, "none"),
    ("res4a_branch1", 1, 512, 1024, "none"),
    ("res4a_branch2a", 1, 512, 256, "relu"),
    ("res4a_branch2b", 3, 256, 256, "relu"),
    ("res4a_branch2c", 1, 256, 1024, "none"),
    ("res4b_branch2a", 1, 1024, 256, "relu"),
    ("res4b_branch2b", 3, 256, 256, "relu"),
    ("res4b_branch2c", 1, 256, 1024, "none"),
    ("res4c_branch2a", 1, 1024, 256, "relu"),
    ("res4c_branch2b", 3, 256, 256, "relu"),
    ("res4c_branch2c", 1, 256, 1024, "none"),
    ("res4d_branch2a", 1, 1024, 256, "relu"),
    ("res4d_branch2b", 3, 256, 256, "relu"),
    ("res4d_branch2c", 1, 256, 1024, "none"),
    ("res4e_branch2a", 1, 1024, 256, "relu"),
    ("res4e_branch2b", 3, 256, 256, "relu"),
    ("res4e_branch2c", 1, 256, 1024, "none"),
    ("res4f_branch2a", 1, 1024, 256, "relu"),
    ("res4f_branch2b", 3, 256, 256, "relu"),
    ("res4f_branch2c", 1, 256, 1024, "none"),
    ("res5a_branch2a_new", 1, 1024, 512, "relu"),
    ("res5a_branch2b_new", 3, 512, 512, "relu"),
    ("res5a_branch2c_new", 1, 512, 1024, "none"),
    ("res5a_branch1_new", 1, 1024, 1024, "none"),
    ("res5b_branch2a_new", 1, 1024, 256, "relu"),
    ("res5b_branch2b_new", 3, 256, 128, "relu"),
    ("res5b_branch2c_new", 1, 128, 256, "relu"),
    ("res5c_branch2b", 3, 212, 128, "relu"),
]

# tf.layers ops without bias: name -> kernel shape
KERNEL_LAYERS = [
    ("res5c_branch1a", (4, 4, 63, 256)),    # conv2d_transpose: (kh, kw, Cout, Cin)
    ("res5c_branch2a", (4, 4, 128, 256)),   # conv2d_transpose
    ("res5c_branch2c", (1, 1, 128, 84)),    # conv2d: (kh, kw, Cin, Cout)
]

BN_SCOPE = "bn5c_branch2a"
BN_CHANNELS = 128


def schema():
    """Ordered list of (name, shape) for all 109 arrays of the reference schema."""
    out = []
    for scope, k, cin, cout, _ in CONV_LAYERS:
        out.append((scope + "/weights", (k, k, cin, cout)))
        out.append((scope + "/biases", (cout,)))
    for scope, shape in KERNEL_LAYERS:
        out.append((scope + "/kernel", shape))
    for leaf in ("gamma", "beta", "moving_mean", "moving_variance"):
        out.append((BN_SCOPE + "/" + leaf, (BN_CHANNELS,)))
    return out


_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64(x):
    """splitmix64 finaliser on a uint64 array (wrap-around arithmetic)."""
    with np.errstate(over="ignore"):
        z = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        return z ^ (z >> np.uint64(31))


def uniform01(seed, n):
    """n floats in [0, 1) with 24 random mantissa bits: element i = hash(hash(seed) + i)."""
    base = splitmix64(np.array([seed], dtype=np.uint64))[0]
    with np.errstate(over="ignore"):
        ctr = base + np.arange(n, dtype=np.uint64)
    bits = splitmix64(ctr) >> np.uint64(40)
    return bits.astype(np.float32) * np.float32(1.0 / (1 << 24))


def _uniform(seed, shape, lo, hi):
    n = int(np.prod(shape))
    u = uniform01(seed, n).astype(np.float64)
    return (lo + (hi - lo) * u).astype(np.float32).reshape(shape)


def synthetic_weights(seed=MASTER_SEED):
    """Seeded weights in the reference schema.

    He-uniform U(+-sqrt(6/fan_in)) for ReLU convs, U(+-sqrt(3/fan_in)) for linear
    convs / transposed convs, biases U(+-0.05), BN gamma U(0.5,1.5), beta and mean
    U(+-0.1), variance U(0.5,1.5).  Array i of the schema uses seed ``seed*1000+i``.
    """
    act = {scope: a for scope, _, _, _, a in CONV_LAYERS}
    w = {}
    for i, (name, shape) in enumerate(schema()):
        s = seed * 1000 + i
        scope, leaf = name.split("/")
        if leaf == "weights":
            fan_in = shape[0] * shape[1] * shape[2]
            b = np.sqrt((6.0 if act[scope] == "relu" else 3.0) / fan_in)
            w[name] = _uniform(s, shape, -b, b)
        elif leaf == "biases":
            w[name] = _uniform(s, shape, -0.05, 0.05)
        elif leaf == "kernel":
            if scope == "res5c_branch2c":
                fan_in = shape[2]
            else:  # transposed conv: every output pixel sums 2x2 taps x Cin
                fan_in = 4 * shape[3]
            b = np.sqrt(3.0 / fan_in)
            w[name] = _uniform(s, shape, -b, b)
        elif leaf in ("gamma", "moving_variance"):
            w[name] = _uniform(s, shape, 0.5, 1.5)
        else:
            w[name] = _uniform(s, shape, -0.1, 0.1)
    return w


def check_schema(weights):
    """Raise ValueError unless `weights` holds every array of the schema with its shape."""
    for name, shape in schema():
        if name not in weights:
            raise ValueError("missing weight array %r" % name)
        got = tuple(np.shape(weights[name]))
        if got != tuple(shape):
            raise ValueError("weight %r has shape %r, expected %r" % (name, got, shape))


def load_weights(path):
    """Read a reference ``params.pkl`` (caffe2pkl.py:83-88 output) or an ``.npz`` with the same keys.

    Prefer ``.npz`` (``np.savez(path, **weights)``): it is plain arrays.  A pickle is what the reference ships between its
    tools (``vnect_model.py:219-222`` unpickles it the same way), but ``pickle.load`` runs arbitrary code from the file: only
    load pickles you made yourself.  Raises ``ValueError`` when an array of the schema is missing or mis-shaped.
    """
    if str(path).endswith(".npz"):
        with np.load(path) as z:
            w = {k: np.asarray(z[k], dtype=np.float32) for k in z.files}
    else:
        with open(path, "rb") as f:
            w = {k: np.asarray(v, dtype=np.float32) for k, v in pickle.load(f).items()}
    check_schema(w)
    return w
