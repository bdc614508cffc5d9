"""Case tables shared by oracle/gen_golden.py and the tests (data only)."""
CONV_CASES = {
    # name: (B, Cin, Cout, H, W, k, stride, padding, dilation, groups, mode)
    "plain": (2, 3, 4, 9, 8, 3, 1, 0, 1, 1, "zeros"),
    "pad": (2, 4, 6, 8, 8, 3, 1, 1, 1, 1, "zeros"),
    "strided": (2, 3, 5, 13, 11, (3, 2), (2, 3), (2, 1), 1, 1, "zeros"),
    "dilated": (1, 2, 3, 12, 12, 3, 1, 2, (3, 2), 1, "zeros"),
    "groups": (2, 4, 6, 7, 7, 3, 1, 1, 1, 2, "zeros"),
    "circular": (2, 3, 4, 8, 9, 3, 1, (2, 1), 1, 1, "circular"),
    "k1": (3, 8, 8, 5, 5, 1, 1, 0, 1, 1, "zeros"),
}

CONV3D_CASES = {
    "base": dict(x=(2, 3, 6, 7, 8), w=(4, 3, 3, 3, 3), kw={}),
    "s2p1": dict(x=(2, 4, 7, 8, 9), w=(6, 4, 3, 2, 3), kw=dict(stride=(2, 1, 2), padding=(1, 0, 1))),
    "dil": dict(x=(1, 2, 9, 9, 9), w=(3, 2, 2, 3, 2), kw=dict(dilation=(2, 1, 3))),
    "groups": dict(x=(2, 4, 5, 6, 6), w=(6, 2, 2, 2, 2), kw=dict(groups=2, padding=1)),
    "circ": dict(x=(2, 2, 5, 6, 7), w=(3, 2, 3, 3, 3), kw=dict(padding=(2, 1, 3), padding_mode="circular")),
}

POOL3D_CASES = {
    "k2": dict(kernel_size=2), "k3s2p1": dict(kernel_size=3, stride=2, padding=1),
    "mixed": dict(kernel_size=(2, 3, 2), stride=(1, 2, 2), padding=(0, 1, 1), dilation=(2, 1, 1)),
    "ceil": dict(kernel_size=(2, 3, 2), stride=(2, 2, 3), ceil_mode=True),
}

CONVT_CASES = {
    # name: x shape, w shape [Cin, Cout / groups, kh, kw], kwargs of cplx.conv_transpose2d
    "base": dict(x=(2, 4, 5, 6), w=(4, 3, 3, 3), kw=dict(groups=1)),
    "s2op1": dict(x=(2, 3, 6, 5), w=(3, 5, 3, 3), kw=dict(stride=2, padding=1, output_padding=1, groups=1)),
    "mixed": dict(x=(1, 4, 7, 4), w=(4, 2, 2, 3), kw=dict(stride=(1, 3), padding=(0, 1), output_padding=(0, 2), groups=1)),
    "dil": dict(x=(2, 2, 5, 5), w=(2, 3, 3, 2), kw=dict(dilation=(2, 3), padding=1, groups=1)),
    "groups": dict(x=(2, 4, 4, 5), w=(4, 3, 3, 3), kw=dict(groups=2, stride=2, padding=1)),
}
