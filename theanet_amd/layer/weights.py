"""Weight initialisation / wrapping.  Mirror of theanet/layer/weights.py: the
draws from the ``numpy.random.RandomState`` seed chain are made in exactly the
reference's order and arithmetic (:50-65) so initial weights are bit-identical."""
import numpy as np

from ..device import DeviceArray, get_context

float_x = 'float32'


def is_shared_var(x):
    """A device-resident array (the analogue of a Theano SharedVariable)."""
    return isinstance(x, DeviceArray)


def borrow(sharedvar, boro=True):
    """Host copy of a device array (weights.py:18-22)."""
    return sharedvar.get_value(borrow=boro)


def init_wb(wb, rand_gen, size_w, size_b, fan_in, fan_out, actvn, name):
    if wb is None or len(wb) == 0:
        if len(size_w) == 4:
            w_values = 2. * rand_gen.randint(2, size=size_w) - 1
            w_values /= np.sqrt(fan_in)
        else:
            w_values = rand_gen.uniform(low=-1, high=1, size=size_w)
            w_values *= np.sqrt(6 / (fan_in + fan_out))

        w_values = np.asarray(w_values, dtype=float_x)
        b_values = np.zeros(size_b, dtype=float_x)

        if actvn == 'sigmoid':
            w_values *= 4
        if actvn in ('softplus', 'relu') or actvn.startswith('relu0'):
            b_values += .5

    elif type(wb[0]) is np.ndarray:
        w_values, b_values = wb[0], wb[1]

    else:
        assert is_shared_var(wb[0])

    if wb is not None and len(wb) and is_shared_var(wb[0]):
        # TestVersion: share the train layer's device buffers
        w, b = wb[0], wb[1]
    else:
        ctx = get_context()
        w = ctx.array(w_values, dtype=float_x)
        b = ctx.array(b_values, dtype=float_x)
        w.name, b.name = name + 'W', name + 'b'
    return w, b
