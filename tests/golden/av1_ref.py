"""Build-container-only helpers: a REAL AV1 encoder and the reference's REAL decoder, both out of Pillow's bundled libavif
(pillow.libs/libavif-*.so = libavif + libaom 3.13.2 [encoder] + dav1d 1.5.3 [decoder]; `from PIL import _avif; _avif.codec_versions()`).

  encode_sequence()   seeded synthetic frames -> libaom (through PIL's AVIF writer) -> the temporal units out of the file's sample table
  Dav1d               ctypes binding of the dav1d entry points the reference calls (src/easyav1.c:1401-1405 dav1d_default_settings /
                      dav1d_open, :2381-2428 dav1d_data_* / dav1d_send_data / dav1d_get_picture, :3059 dav1d_flush, :3969 dav1d_close),
                      which that library exports: OBU bytes in, Dav1dPicture planes out -- the pictures the reference would hand to
                      easyav1_get_video_frame.  dav1d 1.5.3 is the version /root/reference/cmake/dav1d.cmake:18-21 pins.

Nothing here travels to the GPU box as a requirement: fixtures made with it are committed as .npz (tests/golden/make_video_fixtures.py);
bench.py uses Dav1d for the CPU baseline only when the import works there.  Test infrastructure, never on the product path."""
import ctypes as C
import glob
import io
import os
import struct

import numpy as np


def libavif_path():
    import PIL
    hits = glob.glob(os.path.join(os.path.dirname(PIL.__file__), "..", "pillow.libs", "libavif-*.so*"))
    return hits[0] if hits else None


def available():
    try:
        from PIL import _avif
        v = _avif.codec_versions()
        return "dav1d" in v and libavif_path() is not None
    except Exception:
        return False


def versions():
    from PIL import _avif
    return _avif.codec_versions()


# ---------------------------------------------------------------------------------------------------------------- content
def synthetic_frames(seed, w, h, n, kind="natural"):
    """seeded RGB frames (uint8 [h][w][3]): a smooth textured background that pans, two moving rectangles, sensor-like noise;
    kind="screen": flat coloured rectangles and 1-pixel line patterns (palette / intra block copy territory)"""
    rng = np.random.default_rng(seed)
    frames = []
    if kind == "screen":
        cols = rng.integers(0, 256, (6, 3))
        base = np.zeros((h + 64, w + 64, 3), np.uint8)
        base[:] = cols[0]
        for _ in range(14):
            x0, y0 = int(rng.integers(0, w + 32)), int(rng.integers(0, h + 32))
            bw, bh = int(rng.integers(4, w // 2 + 5)), int(rng.integers(4, h // 2 + 5))
            base[y0:y0 + bh, x0:x0 + bw] = cols[rng.integers(1, 6)]
        for _ in range(6):   # "text": alternating 1-pixel columns
            x0, y0 = int(rng.integers(0, w)), int(rng.integers(0, h))
            base[y0:y0 + 7, x0:x0 + 40:2] = cols[rng.integers(1, 6)]
        for i in range(n):
            dx, dy = (i * 3) % 32, (i * 2) % 32
            frames.append(base[dy:dy + h, dx:dx + w].copy())
        return frames
    from PIL import Image
    if kind == "text":    # dark glyphs of a small alphabet on a light page, the same words again and again (intra block copy territory)
        glyphs = rng.integers(0, 2, (24, 7, 5)).astype(np.uint8)
        page = np.full((h + 48, w + 48, 3), 235, np.uint8)
        words = [rng.integers(0, 24, int(rng.integers(3, 9))) for _ in range(12)]
        y = 3
        while y + 8 < h + 48:
            x = 3
            while x + 60 < w + 48:
                for g in words[int(rng.integers(0, 12))]:
                    cell = page[y:y + 7, x:x + 5]
                    cell[glyphs[g] == 1] = (20, 20, 30)
                    x += 6
                x += 5
            y += 10
        for i in range(n):
            frames.append(page[2 * i:2 * i + h, i:i + w].copy())
        return frames
    if kind in ("tiled", "tiled_smooth"):   # large frames: a 320 x 320 texture repeated, panning, with two moving rectangles (cheap to make at 4K)
        t = np.asarray(Image.fromarray(rng.integers(0, 256, (28, 28, 3), dtype=np.uint8)).resize((320, 320), Image.BICUBIC)).astype(np.int16)
        t = np.clip(t + rng.normal(0, 8 if kind == "tiled" else 1.5, t.shape), 0, 255).astype(np.uint8)
        big = np.tile(t, ((h + 96) // 320 + 2, (w + 96) // 320 + 2, 1))
        obj = rng.integers(0, 256, (2, 3)).astype(np.uint8)
        for i in range(n):
            dx, dy = (3 * i) % 320, (2 * i) % 320   # (the texture's period: the pan goes on for any number of frames)
            f = big[dy:dy + h, dx:dx + w].copy()
            for k in range(2):
                ox, oy = (w // 5 + 9 * i * (k + 1)) % (w - 80), (h // 3 + 5 * i * (1 - 2 * k)) % (h - 60)
                f[oy:oy + 48, ox:ox + 64] = obj[k]
            frames.append(f)
        return frames
    if kind == "objects":   # textured triangles / discs with motions of their own over a panning background: oblique moving boundaries (wedge / difference-mask
        # compounds, inter-intra, OBMC territory)
        def texture(th, tw, grain):
            c = rng.integers(0, 256, (th // 10 + 4, tw // 10 + 4, 3), dtype=np.uint8)
            t = np.asarray(Image.fromarray(c).resize((tw, th), Image.BICUBIC)).astype(np.float64)
            return np.clip(t + rng.normal(0, grain, t.shape), 0, 255)
        back = texture(h + 96, w + 96, 5)
        nobj = max(6, w * h // 9000)
        objs = []
        for _ in range(nobj):
            sz = int(rng.integers(24, 90))
            yy, xx = np.mgrid[0:sz, 0:sz]
            shape = int(rng.integers(0, 3))
            if shape == 0:
                m = (xx - sz / 2) ** 2 + (yy - sz / 2) ** 2 < (sz / 2) ** 2
            elif shape == 1:
                m = xx * int(rng.integers(1, 4)) + yy * int(rng.integers(1, 4)) < sz * 2
            else:
                m = (xx - yy * rng.uniform(-0.8, 0.8) > sz * 0.2) & (xx + yy * rng.uniform(0.2, 1.2) < sz * 1.3)
            objs.append(dict(tex=texture(sz, sz, 4), mask=m, x=rng.uniform(0, w), y=rng.uniform(0, h), vx=rng.uniform(-6, 6), vy=rng.uniform(-4, 4), sz=sz))
        for i in range(n):
            dx, dy = (2 * i) % 90, i % 90
            f = back[dy:dy + h, dx:dx + w].copy()
            for o in objs:
                ox, oy = int(round(o["x"] + o["vx"] * i)) % (w + o["sz"]) - o["sz"], int(round(o["y"] + o["vy"] * i)) % (h + o["sz"]) - o["sz"]
                x0, y0, x1, y1 = max(ox, 0), max(oy, 0), min(ox + o["sz"], w), min(oy + o["sz"], h)
                if x1 <= x0 or y1 <= y0:
                    continue
                m = o["mask"][y0 - oy:y1 - oy, x0 - ox:x1 - ox]
                f[y0:y1, x0:x1][m] = o["tex"][y0 - oy:y1 - oy, x0 - ox:x1 - ox][m]
            frames.append(np.clip(f + rng.normal(0, 1.0, f.shape), 0, 255).astype(np.uint8))
        return frames
    if kind == "zoom":    # rotation + zoom about the centre: global / local warped motion territory
        gh, gw = h // 10 + 8, w // 10 + 8
        big = Image.fromarray(rng.integers(0, 256, (gh, gw, 3), dtype=np.uint8)).resize((w + 160, h + 160), Image.BICUBIC)
        big = Image.fromarray(np.clip(np.asarray(big).astype(np.float64) + rng.normal(0, 6, (h + 160, w + 160, 3)), 0, 255).astype(np.uint8))
        for i in range(n):
            r = big.rotate(1.2 * i, resample=Image.BICUBIC)
            z = 1.0 + 0.02 * i
            cw, ch = int(round(w / z)), int(round(h / z))
            x0, y0 = (w + 160 - cw) // 2, (h + 160 - ch) // 2
            frames.append(np.asarray(r.crop((x0, y0, x0 + cw, y0 + ch)).resize((w, h), Image.BICUBIC)).copy())
        return frames
    gh, gw = h // 12 + 6, w // 12 + 6
    coarse = rng.integers(0, 256, (gh, gw, 3)).astype(np.float64)
    big = np.asarray(Image.fromarray(coarse.astype(np.uint8)).resize((w + 96, h + 96), Image.BICUBIC)).astype(np.float64)
    fine = rng.normal(0, 10, big.shape)
    big = np.clip(big + fine, 0, 255)
    obj = rng.integers(0, 256, (2, 3))
    for i in range(n):
        dx, dy = (2 * i + (i * i) // 7) % 90, i % 90
        f = big[dy:dy + h, dx:dx + w].copy()
        for k in range(2):
            ox = (10 + 5 * i * (k + 1)) % max(8, w - 24)
            oy = (h // 3 + 3 * i * (1 - 2 * k)) % max(8, h - 20)
            oh, ow = min(18, h - oy), min(22, w - ox)
            f[oy:oy + oh, ox:ox + ow] = obj[k] + rng.normal(0, 3, (oh, ow, 3))
        f += rng.normal(0, 2.0, f.shape)
        frames.append(np.clip(f, 0, 255).astype(np.uint8))
    return frames


# ---------------------------------------------------------------------------------------------------------------- encode
def _find_box(b, path, off=0, end=None):
    end = len(b) if end is None else end
    while off + 8 <= end:
        sz, typ = struct.unpack(">I4s", b[off:off + 8])
        hdr = 8
        if sz == 1:
            sz = struct.unpack(">Q", b[off + 8:off + 16])[0]
            hdr = 16
        elif sz == 0:
            sz = end - off
        if typ == path[0]:
            if len(path) == 1:
                return off + hdr, off + sz
            return _find_box(b, path[1:], off + hdr, off + sz)
        off += sz
    return None


def samples_of_avif_sequence(data):
    """the samples (one temporal unit each) of the first track of an AVIF image sequence"""
    stbl = [b"moov", b"trak", b"mdia", b"minf", b"stbl"]
    o, _ = _find_box(data, stbl + [b"stsz"])
    _, uniform, n = struct.unpack(">III", data[o:o + 12])
    sizes = [uniform] * n if uniform else list(struct.unpack(">%dI" % n, data[o + 12:o + 12 + 4 * n]))
    co = _find_box(data, stbl + [b"stco"])
    if co:
        o = co[0]
        nc = struct.unpack(">I", data[o + 4:o + 8])[0]
        offs = list(struct.unpack(">%dI" % nc, data[o + 8:o + 8 + 4 * nc]))
    else:
        o, _ = _find_box(data, stbl + [b"co64"])
        nc = struct.unpack(">I", data[o + 4:o + 8])[0]
        offs = list(struct.unpack(">%dQ" % nc, data[o + 8:o + 8 + 8 * nc]))
    o, _ = _find_box(data, stbl + [b"stsc"])
    ne = struct.unpack(">I", data[o + 4:o + 8])[0]
    stsc = [struct.unpack(">III", data[o + 8 + 12 * i:o + 20 + 12 * i]) for i in range(ne)]
    out, k = [], 0
    for ci, off in enumerate(offs):
        per = [e[1] for e in stsc if e[0] <= ci + 1][-1]
        for _ in range(per):
            if k >= n:
                break
            out.append(bytes(data[off:off + sizes[k]]))
            off += sizes[k]
            k += 1
    assert k == n
    return out


def encode_sequence(frames, quality=60, speed=6, subsampling="4:2:0", tile_cols_log2=0, tile_rows_log2=0, advanced=None, threads=1):
    """RGB frames -> libaom -> list of temporal units (bytes, each starting with its temporal delimiter as libaom wrote it).
    `advanced`: dict of libaom key=value options (aom_codec_set_option), e.g. {"enable-cdef": "0", "cq-level": "30"}"""
    from PIL import Image
    ims = [Image.fromarray(np.ascontiguousarray(f)) for f in frames]
    buf = io.BytesIO()
    kw = dict(format="AVIF", codec="aom", subsampling=subsampling, speed=speed, quality=quality, max_threads=threads,
              tile_cols=tile_cols_log2, tile_rows=tile_rows_log2, autotiling=False)
    if advanced:
        kw["advanced"] = {k: str(v) for k, v in advanced.items()}
    if len(ims) > 1:
        ims[0].save(buf, save_all=True, append_images=ims[1:], **kw)
        return samples_of_avif_sequence(buf.getvalue())
    # a still picture: the item's extent in mdat (single item, construction method 0)
    ims[0].save(buf, **kw)
    data = buf.getvalue()
    o, e = _find_box(data, [b"mdat"])
    return [bytes(data[o:e])]


# ---------------------------------------------------------------------------------------------------------------- decode
class Dav1d:
    """dav1d 1.5.3 through the entry points easyav1 calls.  inloop: bit 0 deblock, 1 CDEF, 2 restoration (Dav1dSettings.inloop_filters);
    all_frames: also hand out frames that are not shown (output_invisible_frames)"""

    def __init__(self, threads=1, inloop=7, apply_grain=1, all_frames=False, frame_delay=1):
        self.L = C.CDLL(libavif_path())
        L = self.L
        L.dav1d_version.restype = C.c_char_p
        self.version = L.dav1d_version().decode()
        s = C.create_string_buffer(256)
        L.dav1d_default_settings(s)
        # Dav1dSettings (dav1d 1.x, API 7): n_threads, max_frame_delay, apply_grain, operating_point, all_layers, frame_size_limit, allocator{3 ptr},
        # logger{2 ptr}, strict_std_compliance, output_invisible_frames, inloop_filters, decode_frame_type
        assert struct.unpack_from("i", s, 72)[0] == 7 and struct.unpack_from("i", s, 8)[0] == 1, "unexpected Dav1dSettings layout"
        struct.pack_into("iii", s, 0, threads, frame_delay, apply_grain)
        struct.pack_into("i", s, 16, 0)            # all_layers = 0: one picture per temporal unit like a player wants
        struct.pack_into("ii", s, 68, 1 if all_frames else 0, inloop)
        self.ctx = C.c_void_p()
        rc = L.dav1d_open(C.byref(self.ctx), s)
        if rc:
            raise RuntimeError("dav1d_open: %d" % rc)
        L.dav1d_data_create.restype = C.c_void_p
        self._pic = C.create_string_buffer(512)

    def _take(self, out):
        L = self.L
        pic = C.create_string_buffer(512)
        rc = L.dav1d_get_picture(self.ctx, pic)
        if rc == 0:
            # Dav1dPicture: seq_hdr*, frame_hdr*, data[3], stride[2], p{w, h, layout, bpc}, ...
            _, _, d0, d1, d2, s0, s1, w, h, layout, bpc = struct.unpack_from("5q2q4i", pic, 0)
            assert bpc == 8
            cw, ch = ((w + 1) // 2, (h + 1) // 2) if layout == 1 else ((w + 1) // 2, h) if layout == 2 else (w, h)
            y = np.ctypeslib.as_array(C.cast(d0, C.POINTER(C.c_uint8)), (h, s0))[:, :w].copy()
            if layout == 0:
                u = v = None
            else:
                u = np.ctypeslib.as_array(C.cast(d1, C.POINTER(C.c_uint8)), (ch, s1))[:, :cw].copy()
                v = np.ctypeslib.as_array(C.cast(d2, C.POINTER(C.c_uint8)), (ch, s1))[:, :cw].copy()
            out.append((y, u, v))
            L.dav1d_picture_unref(pic)
        return rc

    def decode(self, tu):
        """one temporal unit -> the pictures it produced"""
        L = self.L
        d = C.create_string_buffer(128)
        p = L.dav1d_data_create(d, C.c_size_t(len(tu)))
        C.memmove(p, tu, len(tu))
        out = []
        while True:
            rc = L.dav1d_send_data(self.ctx, d)
            if rc and rc != -11:     # DAV1D_ERR(EAGAIN)
                L.dav1d_data_unref(d)
                raise RuntimeError("dav1d_send_data: %d" % rc)
            while self._take(out) == 0:
                pass
            if struct.unpack_from("q", d, 8)[0] == 0:
                break
        return out

    def decode_like_easyav1(self, tu):
        """the reference's decode_video loop (src/easyav1.c:2381-2428) on one packet, pictures counted and released without a copy:
        dav1d_send_data, dav1d_get_picture, once more on EAGAIN (which drains the oldest frame in flight), until the packet is consumed"""
        L = self.L
        d = C.create_string_buffer(128)
        p = L.dav1d_data_create(d, C.c_size_t(len(tu)))
        C.memmove(p, tu, len(tu))
        n = 0
        pic = self._pic
        while True:
            rc = L.dav1d_send_data(self.ctx, d)
            if rc and rc != -11:
                L.dav1d_data_unref(d)
                raise RuntimeError("dav1d_send_data: %d" % rc)
            rc = L.dav1d_get_picture(self.ctx, pic)
            if rc == -11:
                rc = L.dav1d_get_picture(self.ctx, pic)
            if rc == 0:
                n += 1
                L.dav1d_picture_unref(pic)
            elif rc != -11:
                raise RuntimeError("dav1d_get_picture: %d" % rc)
            if struct.unpack_from("q", d, 8)[0] == 0:
                return n

    def drain(self):
        out = []
        while self._take(out) == 0:
            pass
        return out

    def flush(self):
        self.L.dav1d_flush(self.ctx)

    def close(self):
        if self.ctx:
            self.L.dav1d_close(C.byref(self.ctx))
            self.ctx = C.c_void_p()


def dav1d_pictures(units, **kw):
    d = Dav1d(**kw)
    out = []
    for tu in units:
        out += d.decode(tu)
    out += d.drain()
    d.close()
    return out
