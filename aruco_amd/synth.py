"""Synthetic frame generator for the BASELINE.json configurations.

Markers carry the reference's 5x5 Hamming bit layout (what FiducidalMarkers::createMarkerImage draws,
/root/reference/src/arucofidmarkers.cpp:217-229: row word = {0x10,0x17,0x09,0x0e}[(id >> 2*(4-y)) & 3], black border
cells) and are rendered through arbitrary homographies with area-sampled (anti-aliased) edges, on a smooth
background with low additive noise. Layout parameters come from a numpy RandomState(seed); rasterisation runs in
torch on whatever device is asked for (the GPU for bench.py, the CPU for the small test cases).
"""
import math

import numpy as np
import torch

_WORDS = (0x10, 0x17, 0x09, 0x0E)


def marker_bits(marker_id: int) -> np.ndarray:
    """7x7 cell matrix (1 = white) of marker `marker_id` including its black border."""
    if not 0 <= marker_id < 1024:
        raise ValueError("marker id out of range")
    m = np.zeros((7, 7), np.uint8)
    for y in range(5):
        val = _WORDS[(marker_id >> 2 * (4 - y)) & 3]
        for x in range(5):
            m[y + 1, x + 1] = (val >> (4 - x)) & 1
    return m


def _homography(src, dst):
    """3x3 H with H*src ~ dst (4 point pairs), float64."""
    A = []
    b = []
    for (x, y), (u, v) in zip(src, dst):
        A.append([x, y, 1, 0, 0, 0, -x * u, -y * u])
        A.append([0, 0, 0, x, y, 1, -x * v, -y * v])
        b += [u, v]
    h = np.linalg.solve(np.array(A, float), np.array(b, float))
    return np.append(h, 1.0).reshape(3, 3)


def _paint_quad(img, quad, table, cells, lo, ss=3):
    """Area-sample a cells x cells table (values in grey levels) mapped onto image quad `quad` (4x2, image px,
    order TL,TR,BR,BL of the table) into float image `img` (H,W). `lo` = table coordinate of the quad's first corner
    (so the quad spans [lo, lo+cells] in table units)."""
    H, W = img.shape
    q = np.asarray(quad, float)
    x0 = max(int(math.floor(q[:, 0].min())) - 1, 0)
    x1 = min(int(math.ceil(q[:, 0].max())) + 2, W)
    y0 = max(int(math.floor(q[:, 1].min())) - 1, 0)
    y1 = min(int(math.ceil(q[:, 1].max())) + 2, H)
    if x1 <= x0 or y1 <= y0:
        return
    src = [(lo, lo), (lo + cells, lo), (lo + cells, lo + cells), (lo, lo + cells)]
    Hinv = _homography([tuple(p) for p in q], src)  # image -> table coords
    dev = img.device
    Hi = torch.tensor(Hinv, dtype=torch.float64, device=dev)
    ys = torch.arange(y0, y1, device=dev, dtype=torch.float64)
    xs = torch.arange(x0, x1, device=dev, dtype=torch.float64)
    tab = torch.as_tensor(table, dtype=torch.float32, device=dev)
    n = tab.shape[0]
    # all ss*ss sub-samples of every pixel at once: pixel (x, y) covers [x-0.5, x+0.5)
    offs = (torch.arange(ss, device=dev, dtype=torch.float64) + 0.5) / ss - 0.5
    px = (xs[None, :] + offs[:, None]).reshape(ss, 1, 1, x1 - x0)   # [ss_x, 1, 1, w]
    py = (ys[None, :] + offs[:, None]).reshape(1, ss, y1 - y0, 1)   # [1, ss_y, h, 1]
    w = Hi[2, 0] * px + Hi[2, 1] * py + Hi[2, 2]
    u = (Hi[0, 0] * px + Hi[0, 1] * py + Hi[0, 2]) / w
    v = (Hi[1, 0] * px + Hi[1, 1] * py + Hi[1, 2]) / w
    inside = (u >= lo) & (u < lo + cells) & (v >= lo) & (v < lo + cells)
    ui = torch.clamp((u - lo).floor().long(), 0, n - 1)
    vi = torch.clamp((v - lo).floor().long(), 0, n - 1)
    val = tab[vi, ui]
    acc = torch.where(inside, val, torch.zeros_like(val)).sum(dim=(0, 1))
    cov = inside.float().sum(dim=(0, 1))
    k = float(ss * ss)
    region = img[y0:y1, x0:x1]
    img[y0:y1, x0:x1] = region * (1 - cov / k) + acc / k


def _marker_table(marker_id, quiet, black, white):
    n = 7 + 2 * quiet
    t = np.full((n, n), float(white), np.float32)
    bits = marker_bits(marker_id)
    t[quiet:quiet + 7, quiet:quiet + 7] = np.where(bits > 0, float(white), float(black))
    return t


def _poly_mask(shape, quad, scale):
    """Occupancy of a convex quad on a grid downscaled by `scale`: (y0, y1, x0, x1, mask of that window)."""
    h, w = shape
    q = np.asarray(quad, float) / scale
    x0 = int(max(0, math.floor(q[:, 0].min()) - 1)); x1 = int(min(w, math.ceil(q[:, 0].max()) + 2))
    y0 = int(max(0, math.floor(q[:, 1].min()) - 1)); y1 = int(min(h, math.ceil(q[:, 1].max()) + 2))
    if x1 <= x0 or y1 <= y0:
        return 0, 0, 0, 0, np.zeros((0, 0), bool)
    ys, xs = np.mgrid[y0:y1, x0:x1]
    m = np.ones((y1 - y0, x1 - x0), bool)
    sign = 0
    for i in range(4):
        ax, ay = q[i]
        bx, by = q[(i + 1) % 4]
        cr = (bx - ax) * (ys + 0.5 - ay) - (by - ay) * (xs + 0.5 - ax)
        if sign == 0:
            c = (q[:, 0].mean() - ax) * (by - ay) - (q[:, 1].mean() - ay) * (bx - ax)
            sign = -1 if c > 0 else 1
        m &= (cr * sign) >= 0
    return y0, y1, x0, x1, m


def frame_layout(rng, width, height, n_markers=20, side_range=(90, 220), margin=60, jitter=0.08, quiet=1, cells=7, id_pool=1024):
    """Random non-overlapping marker placement. Returns list of dicts(id, quad (4x2 TL,TR,BR,BL of the 7x7 marker),
    quad_q (quiet-zone quad))."""
    ids = rng.choice(id_pool, size=n_markers, replace=False)   # cells = marker cells per side incl. the black border
    sc = 4
    occ = np.zeros(((height + sc - 1) // sc, (width + sc - 1) // sc), bool)
    out = []
    for mid in ids:
        for attempt in range(200):
            hi = side_range[1] - (side_range[1] - side_range[0]) * min(attempt / 60.0, 0.95)
            s = rng.uniform(side_range[0], max(hi, side_range[0] + 1))
            th = rng.uniform(0, 2 * math.pi)
            c, sn = abs(math.cos(th)), abs(math.sin(th))
            if s * max(c, sn) < 86:  # keeps the border contour above the 0.04*max(W,H)*4 size filter at 1080p
                continue
            cx = rng.uniform(margin, width - margin)
            cy = rng.uniform(margin, height - margin)
            R = np.array([[math.cos(th), -math.sin(th)], [math.sin(th), math.cos(th)]])
            base = np.array([[-0.5, -0.5], [0.5, -0.5], [0.5, 0.5], [-0.5, 0.5]]) * s
            base = base + rng.uniform(-jitter, jitter, size=(4, 2)) * s  # mild perspective
            quad = base @ R.T + np.array([cx, cy])
            # quiet-zone quad: extend the marker's own projective frame by `quiet` cells
            Hm = _homography([(0, 0), (cells, 0), (cells, cells), (0, cells)], [tuple(p) for p in quad])
            qq = []
            for (u, v) in [(-quiet, -quiet), (cells + quiet, -quiet), (cells + quiet, cells + quiet), (-quiet, cells + quiet)]:
                p = Hm @ np.array([u, v, 1.0])
                qq.append(p[:2] / p[2])
            qq = np.array(qq)
            if quad[:, 0].min() < margin or quad[:, 0].max() > width - margin or quad[:, 1].min() < margin or \
                    quad[:, 1].max() > height - margin:
                continue
            if qq[:, 0].min() < 2 or qq[:, 0].max() > width - 3 or qq[:, 1].min() < 2 or qq[:, 1].max() > height - 3:
                continue
            # is the quad convex / well formed?
            d = np.roll(qq, -1, axis=0) - qq
            cr = d[:, 0] * np.roll(d, -1, axis=0)[:, 1] - d[:, 1] * np.roll(d, -1, axis=0)[:, 0]
            if not (np.all(cr > 0) or np.all(cr < 0)):
                continue
            # pad the occupancy test by a few pixels so quiet zones never touch
            ctr = qq.mean(axis=0)
            grown = ctr + (qq - ctr) * (1 + 10.0 / s)
            y0, y1, x0, x1, m = _poly_mask(occ.shape, grown, sc)
            if (m & occ[y0:y1, x0:x1]).any():
                continue
            occ[y0:y1, x0:x1] |= m
            out.append({"id": int(mid), "quad": quad, "quad_q": qq})
            break
    return out


def render_frame(layout, width, height, rng, device="cpu", noise_sigma=1.5, quiet=1, gen=None, cells=7, table_fn=None, clutter=False):
    """Rasterise one frame (uint8 tensor HxW on `device`) for a layout from frame_layout(). clutter: a two-level blob texture
    (cell size ~24 px, smooth outlines) behind the markers instead of the flat background — hundreds of long borders per frame that
    are followed, approximated and rejected (the robustness leg of bench.py; the parity tests compare such frames with the CPU restatement)."""
    dev = torch.device(device)
    base = rng.uniform(150, 210)
    gx, gy = rng.uniform(-12, 12, size=2)
    ys = torch.linspace(-0.5, 0.5, height, device=dev)[:, None]
    xs = torch.linspace(-0.5, 0.5, width, device=dev)[None, :]
    img = (base + gx * xs + gy * ys).to(torch.float32).expand(height, width).contiguous()
    if clutter:
        g2 = torch.Generator(device="cpu")
        g2.manual_seed(int(rng.randint(0, 2 ** 31 - 1)))
        field = torch.randn((1, 1, height // 24 + 3, width // 24 + 3), generator=g2).to(dev)
        field = torch.nn.functional.interpolate(field, size=(height, width), mode="bicubic", align_corners=False)[0, 0]
        img = torch.where(field > 0.25, img - rng.uniform(70, 110), img)
    for mk in layout:
        black = rng.uniform(15, 45)
        white = rng.uniform(215, 245)
        t = (table_fn or _marker_table)(mk["id"], quiet, black, white)
        _paint_quad(img, mk["quad_q"], t, cells + 2 * quiet, -quiet)
    if noise_sigma > 0:
        if gen is None:
            gen = torch.Generator(device=dev)
            gen.manual_seed(int(rng.randint(0, 2 ** 31 - 1)))
        img = img + torch.randn(img.shape, generator=gen, device=dev) * noise_sigma
    return img.round().clamp(0, 255).to(torch.uint8)


def make_stream(n_frames, width=1920, height=1080, seed=4711, n_markers=20, device="cpu", noise_sigma=1.5, clutter=False):
    """Config-2/3/5 stream: returns (frames uint8 [N,H,W] on device, truth list per frame)."""
    rng = np.random.RandomState(seed)
    frames = torch.empty((n_frames, height, width), dtype=torch.uint8, device=device)
    truth = []
    for f in range(n_frames):
        lay = frame_layout(rng, width, height, n_markers=n_markers,
                           side_range=(90 * max(width, height) / 1920.0, 220 * max(width, height) / 1920.0),
                           margin=int(60 * max(width, height) / 1920.0))
        frames[f] = render_frame(lay, width, height, rng, device=device, noise_sigma=noise_sigma, clutter=clutter)
        truth.append(lay)
    return frames, truth


def make_hrm_frame(markers, width=1280, height=720, seed=7, n_markers=12, device="cpu", noise_sigma=1.5):
    """One frame with highly reliable markers: `markers` = the dictionary's bit strings (n*n characters, '1' = white cell),
    drawn like MarkerCode::getImg (src/highlyreliablemarkers.cpp:238-260: n x n code inside a one-cell black border).
    Returns (uint8 frame [H,W], layout with id = position in the dictionary)."""
    n = int(round(len(markers[0]) ** 0.5))
    cells = n + 2
    rng = np.random.RandomState(seed)
    scale = max(width, height) / 1920.0
    lay = frame_layout(rng, width, height, n_markers=min(n_markers, len(markers)), side_range=(110 * scale * 1.5, 220 * scale * 1.5),
                       margin=int(60 * scale), cells=cells, id_pool=len(markers))

    def table(marker_id, quiet, black, white):
        t = np.full((cells + 2 * quiet, cells + 2 * quiet), float(white), np.float32)
        t[quiet:quiet + cells, quiet:quiet + cells] = float(black)
        bits = np.array([c == "1" for c in markers[marker_id]]).reshape(n, n)
        t[quiet + 1:quiet + 1 + n, quiet + 1:quiet + 1 + n] = np.where(bits, float(white), float(black))
        return t

    return render_frame(lay, width, height, rng, device=device, noise_sigma=noise_sigma, cells=cells, table_fn=table), lay


# ---------------------------------------------------------------------------------------------
# Board frames (config 4): the 6x4 layout of testdata/board/board_pix.yml seen through a pinhole camera
# ---------------------------------------------------------------------------------------------
def _rodrigues(r):
    th = np.linalg.norm(r)
    if th < 1e-12:
        return np.eye(3)
    k = r / th
    Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + math.sin(th) * Kx + (1 - math.cos(th)) * (Kx @ Kx)


def project(K, rvec, tvec, pts3):
    R = _rodrigues(np.asarray(rvec, float))
    p = (R @ np.asarray(pts3, float).T).T + np.asarray(tvec, float)
    uv = p[:, :2] / p[:, 2:3]
    return np.stack([uv[:, 0] * K[0, 0] + K[0, 2], uv[:, 1] * K[1, 1] + K[1, 2]], axis=1)


def render_board(ids, obj, K, rvec, tvec, width, height, rng, device="cpu", noise_sigma=1.5, unit=1.0, pad=40.0):
    """ids[N], obj[N][4][3] in board units (scaled by `unit` to metres). Returns uint8 frame and the projected
    marker quads."""
    dev = torch.device(device)
    obj = np.asarray(obj, float) * unit
    base = rng.uniform(90, 130)
    img = torch.full((height, width), float(base), dtype=torch.float32, device=dev)
    lo = obj.reshape(-1, 3).min(axis=0) - pad * unit
    hi = obj.reshape(-1, 3).max(axis=0) + pad * unit
    sheet = np.array([[lo[0], lo[1], 0], [hi[0], lo[1], 0], [hi[0], hi[1], 0], [lo[0], hi[1], 0]])
    white = rng.uniform(215, 240)
    _paint_quad(img, project(K, rvec, tvec, sheet), np.full((1, 1), white, np.float32), 1, 0)
    quads = []
    for mid, o in zip(ids, obj):
        q = project(K, rvec, tvec, o)
        quads.append(q)
        t = _marker_table(int(mid), 0, rng.uniform(15, 40), white)
        _paint_quad(img, q, t, 7, 0)
    if noise_sigma > 0:
        gen = torch.Generator(device=dev)
        gen.manual_seed(int(rng.randint(0, 2 ** 31 - 1)))
        img = img + torch.randn(img.shape, generator=gen, device=dev) * noise_sigma
    return img.round().clamp(0, 255).to(torch.uint8), quads


def make_board_stream(n_frames, ids, obj, K, width=3840, height=2160, seed=4711, device="cpu", unit=0.039 / 100.0,
                      z_range=(0.52, 0.60), noise_sigma=1.5):
    """Config-4 stream: the board (ids, obj[N][4][3] in board units) seen through random poses (board x right, y down,
    z away from the camera, as testdata/board/board_pix.yml defines it). Returns (frames uint8 [N,H,W], poses)."""
    rng = np.random.RandomState(seed)
    frames = torch.empty((n_frames, height, width), dtype=torch.uint8, device=device)
    poses = []
    for f in range(n_frames):
        rvec = rng.uniform(-0.2, 0.2, 3)
        tvec = np.array([rng.uniform(-0.02, 0.02), rng.uniform(-0.02, 0.02), rng.uniform(*z_range)])
        fr, _ = render_board(ids, obj, np.asarray(K, float).reshape(3, 3), rvec, tvec, width, height, rng, device=device,
                             noise_sigma=noise_sigma, unit=unit)
        frames[f] = fr
        poses.append((rvec, tvec))
    return frames, poses
