"""Algorithmic FLOP / byte counts of the hot-path kernels: what
``bench.py`` divides measured kernel time into.  fp32 everywhere (4 B/element).

FLOPs: 2*M*N*K per GEMM-shaped op.  Bytes: every input read once + every output
written once, no im2col buffer, no re-reads.
"""

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec (MI355X_MICROARCH.md: 6.3 TB/s achievable)
MFMA_F32_PEAK_TFLOPS = 157.3  # v_mfma_f32_32x32x2_f32 dense peak
MFMA_F16_PEAK_TFLOPS = 2500.0  # v_mfma_f32_32x32x16_f16 dense peak (MI355X_MICROARCH.md; not the 2:1-sparse figure)


def conv_shapes(N, C, H, K, f, Ho):
    return dict(M=N * Ho * Ho, N=K, K=C * f * f,
                x_bytes=4 * N * C * H * H, a_bytes=4 * N * K * Ho * Ho, w_bytes=4 * K * C * f * f)


def kernel_cost(kind, **d):
    """(flops, algorithmic_bytes) of one launch."""
    if kind == "conv_fwd":
        s = conv_shapes(**d)
        return 2 * s["M"] * s["N"] * s["K"], s["x_bytes"] + s["w_bytes"] + s["a_bytes"]
    if kind == "conv_wgrad":
        s = conv_shapes(**d)
        return 2 * s["M"] * s["N"] * s["K"], s["x_bytes"] + s["a_bytes"] + s["w_bytes"]
    if kind == "conv_dgrad":
        s = conv_shapes(**d)
        return 2 * s["M"] * s["N"] * s["K"], s["a_bytes"] + s["w_bytes"] + s["x_bytes"]
    if kind in ("fc_fwd", "fc_dgrad", "fc_wgrad"):
        B, n_in, n_out = d["B"], d["n_in"], d["n_out"]
        byt = 4 * (B * n_in + n_in * n_out + B * n_out)
        return 2 * B * n_in * n_out, byt
    if kind == "pool_fwd":
        return 0, 4 * d["NC"] * (d["H"] ** 2 + d["Ho"] ** 2)
    if kind == "pool_bwd":   # reads x, y, dy ; writes dx
        return 0, 4 * d["NC"] * (2 * d["H"] ** 2 + 2 * d["Ho"] ** 2)
    if kind == "elastic_apply":
        return 0, 8 * d["N"] * d["C"] * d["h"] * d["w"]
    raise KeyError(kind)


def net_step_flops(net):
    """Algorithmic FLOPs of one training step (sum of 2MNK x 3, x 2 for the
    first parametrised layer, which needs no dgrad); per local batch."""
    from .layer import ConvLayer, HiddenLayer
    total, first = 0, True
    for lyr in net.tr_layers:
        if isinstance(lyr, ConvLayer):
            f = 2 * lyr.batch_sz * lyr.out_sz ** 2 * lyr.num_maps * lyr.num_prev_maps * lyr.filter_sz ** 2
        elif isinstance(lyr, HiddenLayer):
            f = 2 * lyr.batch_sz * lyr.n_in * lyr.n_out
        else:
            continue
        total += f * (2 if first else 3)
        first = False
    return total
