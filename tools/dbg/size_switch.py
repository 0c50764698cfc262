import sys
sys.path.insert(0, '/root/repo')
import numpy as np
import torch
torch.zeros(1, device="cuda")
from tests import test_real_av1 as T
from tests import webm_writer as ww
from easyav1_amd import easyav1 as ea
ea.api()

def units_of(fx):
    out = []
    for tu in fx["units"]:
        body = b""
        for typ, off, size in T.split_obus(tu):
            if typ == 2:
                continue
            body += tu[off:off + size]
        out.append((body, any(typ == 1 for typ, off, size in T.split_obus(tu))))
    return out

def first_seq(fx):
    for tu in fx["units"]:
        for typ, off, size in T.split_obus(tu):
            if typ == 1:
                return tu[off:off + size]

def mux(fxs, track_of=0):
    dur = 40
    blocks, k = [], 0
    for fx in fxs:
        for body, key in units_of(fx):
            blocks.append(dict(track=1, tc=dur * k, key=key, frames=[body], lacing=0, group=False, duration=None))
            k += 1
    f0 = fxs[track_of]
    track = dict(number=1, type=1, codec="V_AV1", private=bytes([0x81, 0x00, 0x0C, 0x00]) + first_seq(fxs[0]), default_duration=dur * 1000000, width=f0["width"], height=f0["height"])
    return ww.make_webm(11, [track], blocks, duration=float(k * dur), extras=False)

def run(names, track_of):
    fxs = [T.load_fixture([p for p in T.ALL_FIXTURES if p.endswith("av1_stream_%s.npz" % n)][0]) for n in names]
    want = [(fx, i) for fx in fxs for i in range(len(fx["shown_sha"]))]
    p = ea.Player(mux(fxs, track_of), enable_audio=0, skip_unprocessed_frames=0, log_level=ea.LOG_ERROR)
    shown, bad = 0, 0
    while p.decode_next() == ea.STATUS_OK:
        if not p.has_video_frame():
            continue
        ts, planes, props = p.video_frame()
        fx, i = want[shown]
        ok = planes[0].shape == (fx["height"], fx["width"]) and np.array_equal(T.sha([x for x in planes if x is not None], fx["layout"] == 0), fx["shown_sha"][i])
        bad += not ok
        if not ok and bad < 4:
            print("  picture", shown, "shape", planes[0].shape, "expected", (fx["height"], fx["width"]))
        shown += 1
    print(names, "track of", track_of, ": shown", shown, "of", len(want), "wrong", bad, "status", p.get_status())
    p.close()

run(["deltalf_q30_264x200", "pyramid_200x136"], 0)
run(["pyramid_200x136", "deltalf_q30_264x200"], 1)
run(["pyramid_200x136", "deltalf_q30_264x200"], 0)
run(["pyramid_200x136", "c444_screen_200x136"], 0)
run(["pyramid_200x136", "mono_136x104"], 0)
