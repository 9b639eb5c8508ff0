"""The caller's side of the path: Tracking::GrabImage* converts colour input with cv::cvtColor(... 2GRAY) before ExtractORB
(reference src/Tracking.cc:915-941, 985-1001).  CPU: the oracle against the published fixed-point formula and its
properties; GPU: orbx_gray_from_color_device against the oracle, bit-exact, for every channel order / alignment."""
import numpy as np
import pytest

import oracle_lib as O
import extractorb_amd as X


def formula(img, red_first):
    a = img.astype(np.int64)
    r, g, b = (a[..., 0], a[..., 1], a[..., 2]) if red_first else (a[..., 2], a[..., 1], a[..., 0])
    return ((r * 4899 + g * 9617 + b * 1868 + 8192) >> 14).astype(np.uint8)


@pytest.mark.parametrize("channels", [3, 4])
@pytest.mark.parametrize("red_first", [True, False])
def test_oracle_gray_is_the_14_bit_fixed_point_formula(channels, red_first):
    rng = np.random.default_rng(channels * 2 + red_first)
    img = rng.integers(0, 256, (37, 53, channels), dtype=np.uint8)
    img[0, :8] = [[0] * channels, [255] * channels, [255, 0, 0, 9][:channels], [0, 255, 0, 9][:channels], [0, 0, 255, 9][:channels],
                  [1, 1, 1, 0][:channels], [254, 255, 254, 0][:channels], [128, 127, 129, 255][:channels]]
    got = O.gray_from_color(img, red_first)
    assert np.array_equal(got, formula(img, red_first))
    assert got[0, 0] == 0 and got[0, 1] == 255                      # the coefficients sum to 2^14: white stays 255
    gray3 = np.repeat(rng.integers(0, 256, (5, 7, 1), dtype=np.uint8), channels, axis=2)
    assert np.array_equal(O.gray_from_color(gray3, red_first), gray3[..., 0])     # a gray colour image is a fixed point
    if channels == 4:                                               # alpha is ignored
        img2 = img.copy(); img2[..., 3] = 255 - img2[..., 3]
        assert np.array_equal(O.gray_from_color(img2, red_first), got)
    assert np.array_equal(O.gray_from_color(img[..., [2, 1, 0, 3][:channels]], not red_first), got)   # RGB vs BGR


@pytest.mark.gpu
@pytest.mark.parametrize("channels,red_first,cols,pad", [(3, True, 640, 0), (3, False, 641, 0), (4, True, 640, 0), (4, False, 322, 8),
                                                         (3, True, 37, 5), (3, False, 640, 3)])
def test_gpu_gray_equals_oracle_and_feeds_the_extractor(channels, red_first, cols, pad):
    import torch
    rows, B = 96 if cols < 100 else 240, 3
    rng = np.random.default_rng(cols + channels)
    stride = cols * channels + pad                                  # pad 3 / 5: rows that are not dword aligned
    src = rng.integers(0, 256, (B, rows, stride), dtype=np.uint8)
    d_src = torch.from_numpy(src).cuda()
    d_gray = torch.zeros((B, rows, cols), dtype=torch.uint8, device="cuda")
    ex = X.ORBextractor(300, max_width=max(cols, 320), max_height=max(rows, 240), max_batch=B)
    ex.set_stream(torch.cuda.current_stream().cuda_stream)
    ex.gray_from_color_device(B, d_src, rows, cols, channels, red_first, d_gray, src_stride=stride)
    torch.cuda.synchronize()
    got = d_gray.cpu().numpy()
    for f in range(B):
        img = src[f, :, :cols * channels].reshape(rows, cols, channels)
        assert np.array_equal(got[f], O.gray_from_color(img, red_first)), "frame %d" % f
    if cols >= 320:                                                 # and the gray frames go straight into the extractor
        cap = ex.capacity
        d_k = torch.zeros((B, cap, 7), dtype=torch.float32, device="cuda"); d_d = torch.zeros((B, cap, 32), dtype=torch.uint8, device="cuda")
        d_n = torch.zeros(B, dtype=torch.int32, device="cuda"); d_m = torch.zeros(B, dtype=torch.int32, device="cuda")
        ex.extract_batch_device(d_gray, B, rows, cols, d_k, d_d, d_n, d_m, cap)
        torch.cuda.synchronize()
        one = X.ORBextractor(300, max_width=cols, max_height=rows)
        mono, k, d, _ = one(got[1])
        assert int(d_n[1]) == len(k) and np.array_equal(d_d[1, :len(k)].cpu().numpy(), d)


@pytest.mark.gpu
def test_gpu_gray_argument_errors():
    ex = X.ORBextractor(300)
    with pytest.raises(X.OrbxError):
        ex.gray_from_color_device(1, 1, 10, 10, 2, True, 1)
    with pytest.raises(X.OrbxError):
        ex.gray_from_color_device(1, 1, 10, 10, 3, True, 1, src_stride=20)
