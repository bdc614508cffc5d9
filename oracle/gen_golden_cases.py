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
