"""Deterministic synthetic inputs (host side, no RNG).

`dynamic_test_image` follows the recipe of the reference's Data/dynamicTestImage.m:17-55
(ellipse with alternating sin^2 textures, frame and rectangle, Gaussian-blurred with
sigma = (m+n)/150).  The Image Processing Toolbox blur `imgaussfilt` is replaced by an own
definition: separable Gaussian, radius ceil(2*sigma), replicate padding, taps normalised to 1.
Runs on any torch device (the 4096^2 x 16 benchmark frames are generated on the GPU).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F


def gaussian_blur(img, sigma):
    """img: (m, n) float64 tensor."""
    r = int(math.ceil(2 * sigma))
    t = torch.arange(-r, r + 1, dtype=torch.float64, device=img.device)
    w = torch.exp(-(t * t) / (2.0 * sigma * sigma))
    w = w / w.sum()
    x = img[None, None]
    x = F.pad(x, (0, 0, r, r), mode="replicate")
    x = F.conv2d(x, w.view(1, 1, -1, 1))
    x = F.pad(x, (r, r, 0, 0), mode="replicate")
    x = F.conv2d(x, w.view(1, 1, 1, -1))
    return x[0, 0]


def dynamic_test_image(m=200, n=200, num_frames=6, device="cpu"):
    """Returns a (num_frames, m, n) float64 tensor of frames normalised to [0, 1] each
    (the drivers normalise per image, NNE/mf_nn_tv_registration_no_ref_ml.m:49-51)."""
    dev = torch.device(device)
    xs = torch.linspace(-1, 1, n, dtype=torch.float64, device=dev)
    ys = torch.linspace(-1, 1, m, dtype=torch.float64, device=dev)
    yy, xx = torch.meshgrid(ys, xs, indexing="ij")
    f1, p1 = 4 * math.pi, 0.375 * math.pi
    f2, p2 = 6 * math.pi, 0.125 * math.pi
    out = torch.empty((num_frames, m, n), dtype=torch.float64, device=dev)
    for i in range(1, num_frames + 1):
        dx = -0.1 * math.sin(math.pi * i / num_frames)
        dy = 0.1 * math.cos(math.pi * i / num_frames)
        rad = 0.4
        rr = torch.sqrt(2 * (xx + dx - 0.25) ** 2 + (yy + dy) ** 2)
        ellipse = (rr <= rad).to(torch.float64)
        if i % 2 == 0:
            texture = torch.sin(f1 * (yy + dy) + p1) ** 2
        else:
            texture = torch.sin(f2 * (xx + dx) + p2) ** 2
        idx = rr <= 0.6 * rad
        ellipse = torch.where(idx, texture, ellipse)
        mx = torch.maximum(xx.abs(), yy.abs())
        frame = ((0.7 <= mx) & (mx <= 0.85)).to(torch.float64)
        rect = ((-0.6 <= xx) & (xx <= -0.2) & (-0.5 <= yy) & (yy <= 0.5)).to(torch.float64)
        t = gaussian_blur(rect + ellipse + frame, (m + n) / 150.0)
        out[i - 1] = (t - t.min()) / (t.max() - t.min())
    return out


def dynamic_test_landmarks(m=200, n=200, num_frames=6):
    """the second output of Data/dynamicTestImage.m: LM (17, 2, num_frames) in the coordinate system
    omega = [0, m] x [0, n] (:57-81): four corners of the rectangle, five points that move with the ellipse, eight fixed
    points on the frame. (The file allocates 16 rows, :19, and assigns 17; MATLAB grows the array.)"""
    LM = np.zeros((17, 2, num_frames))
    r2 = 0.4 / math.sqrt(2.0)
    for i in range(1, num_frames + 1):
        dx = -0.1 * math.sin(math.pi * i / num_frames)
        dy = 0.1 * math.cos(math.pi * i / num_frames)
        LM[:, :, i - 1] = [[-0.5, -0.6], [-0.5, -0.2], [0.5, -0.6], [0.5, -0.2],
                           [-dy, r2 - (dx - 0.25)], [-dy, -r2 - (dx - 0.25)],
                           [0.4 - dy, 0.25 - dx], [-0.4 - dy, 0.25 - dx], [-dy, 0.25 - dx],
                           [0.7, 0.7], [0.7, -0.7], [-0.7, 0.7], [-0.7, -0.7],
                           [0.85, 0.85], [0.85, -0.85], [-0.85, 0.85], [-0.85, -0.85]]
    LM = LM + 1.0
    LM[:, 0, :] *= m / 2.0
    LM[:, 1, :] *= n / 2.0
    return LM


def smooth_displacement(m, n, k, amp=2.0, device="cpu"):
    """Deterministic smooth start displacement u0 ~ (m*n, 2, k) (column-major pixel order):
    u_x = amp*sin(2 pi i/m) cos(2 pi j/n), u_y = amp*cos(2 pi i/m) sin(2 pi j/n), scaled per image."""
    dev = torch.device(device)
    i = (torch.arange(m, dtype=torch.float64, device=dev) + 0.5) / m
    j = (torch.arange(n, dtype=torch.float64, device=dev) + 0.5) / n
    ux = amp * torch.sin(2 * math.pi * i)[:, None] * torch.cos(2 * math.pi * j)[None, :]
    uy = amp * torch.cos(2 * math.pi * i)[:, None] * torch.sin(2 * math.pi * j)[None, :]
    out = torch.empty((k, 2, n, m), dtype=torch.float64, device=dev)   # [img][comp][j][i]
    for q in range(k):
        s = math.cos(math.pi * q / max(k, 1))
        out[q, 0] = (s * ux).T
        out[q, 1] = (s * uy).T
    return out


def shifted_rect_pair(m, n, shift):
    """A 'sliding_rect'-like pair: blurred rectangle and its copy shifted by integer pixels."""
    R = np.zeros((m, n))
    R[m // 4: m // 4 + m // 3, n // 4: n // 4 + n // 3] = 1.0
    T = np.roll(np.roll(R, shift[0], axis=0), shift[1], axis=1)
    k = np.array([0.25, 0.5, 0.25])
    for a in (R, T):
        for _ in range(2):
            a[1:-1, :] = k[0] * a[:-2, :] + k[1] * a[1:-1, :] + k[2] * a[2:, :]
            a[:, 1:-1] = k[0] * a[:, :-2] + k[1] * a[:, 1:-1] + k[2] * a[:, 2:]
    return R, T
