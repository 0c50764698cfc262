"""WebM file -> packets -> OBU layer -> frames with headers and tile byte ranges; audio packets of the same
file -> Vorbis front end.  Everything the host does before the AV1 symbol decode, through the C ABI of
this repo only (the file comes from the test-side writers: no muxer / encoder exists here)."""
import numpy as np
import pytest

from tests import av1_frame_header_writer as fw
from tests import av1_obu_writer as wr
from tests import test_frame_header as tfh
from tests import vorbis_stream_gen as vgen
from tests import webm_writer as ww


@pytest.fixture(scope="module")
def libs():
    import easyav1_amd
    easyav1_amd.load(build=True)
    from easyav1_amd import obu, webm
    return obu, webm


def test_webm_av1_track_to_frames(libs):
    obu, webm = libs
    rng = np.random.default_rng(2025)
    seq = tfh.simple_seq(max_width=640, max_height=360, film_grain_params_present=1)
    seq_obu = wr.obu(1, wr.write_sequence_header(seq))
    slots = fw.Slots()
    vhdr = vgen.make_stream(3, channels=2)
    apk = vgen.random_packets(3, vhdr[3], count=20)
    tracks = [dict(number=1, type=1, codec="V_AV1", private=bytes([0x81, 0x00, 0x0C, 0x00]) + seq_obu, default_duration=33366700, width=640, height=360),
              dict(number=2, type=2, codec="A_VORBIS", private=ww.vorbis_private(list(vhdr[:3])), rate=48000.0, channels=2)]
    blocks, expect, tc = [], [], 0
    for frame_no in range(12):
        have_refs = all(s0 is not None for s0 in slots.s)
        ft = tfh.KEY if not have_refs else int(rng.choice([tfh.KEY, tfh.INTER, tfh.INTER, tfh.INTER]))
        h = tfh.plain_header(rng, seq, slots, ft, frame_no, show_frame=1)
        hb, nbits = fw.write_frame_header(seq, h, slots)
        ntiles = h["tile_cols"] * h["tile_rows"]
        bodies = [bytes(rng.integers(0, 256, int(rng.integers(1, 60)), dtype=np.uint8)) for _ in range(ntiles)]
        w = fw.Bits()
        if ntiles > 1:
            w.f(1, 0)
        while len(w.bits) % 8:
            w.bits.append(0)
        tg = w.bytes()
        for t in range(ntiles):
            if t != ntiles - 1:
                tg += (len(bodies[t]) - 1).to_bytes(h["tile_size_bytes"], "little")
            tg += bodies[t]
        tu = wr.obu(2, b"") + (seq_obu if ft == tfh.KEY else b"") + wr.obu(6, hb + tg)
        blocks.append(dict(track=1, tc=tc, key=ft == tfh.KEY, frames=[tu], lacing=0, group=False, duration=None))
        if frame_no < len(apk):
            blocks.append(dict(track=2, tc=tc, key=True, frames=[apk[frame_no]], lacing=0, group=False, duration=None))
        expect.append((h, bodies, tc))
        slots.refresh(h)
        tc += 33
    data = ww.make_webm(11, tracks, blocks, duration=float(tc))
    r = webm.WebmReader(data)
    assert [t.codec for t in r.tracks] == [4, 1]        # EA1_WEBM_CODEC_AV1, EA1_WEBM_CODEC_VORBIS
    # the sequence header can be found in the codec private data (av1C: 4 bytes, then configOBUs) and in the key frames
    rc, sh = obu.parse_sequence_header(r.codec_data(0, 0)[4:])
    assert rc == 0 and (sh["max_width"], sh["max_height"], sh["layout"], sh["hbd"]) == (640, 360, 1, 0)
    parser = obu.Parser()
    frames = audio = 0
    from easyav1_amd import vorbis
    dec = vorbis.VorbisDecoder(r.codec_data(1, 0), r.codec_data(1, 1), r.codec_data(1, 2), device=-1)
    while True:
        p = r.read_packet()
        if p is None:
            break
        if p["track"] == 1:
            spectra, long_block = dec.spectra(p["frames"][0])
            assert spectra is None or spectra.shape[0] == 2
            audio += 1
            continue
        h, bodies, tc = expect[frames]
        assert p["tstamp"] == tc * 1000000 and p["keyframe"] == int(h["frame_type"] == tfh.KEY)
        pkt = p["frames"][0]
        rc, used, got, tiles = obu.next_frame(parser, pkt)
        assert rc == 0 and used == len(pkt)
        d = got.as_dict()
        assert (d["frame_type"], d["frame_width"], d["frame_height"], d["refresh_frame_flags"]) == (h["frame_type"], h["frame_width"], h["frame_height"],
                                                                                                     h["refresh_frame_flags"])
        assert [pkt[t["offset"]: t["offset"] + t["size"]] for t in tiles] == bodies
        parser.end_frame(got)
        frames += 1
    assert frames == len(expect) and audio == min(len(apk), 12)
    dec.close()
    parser.close()
    r.close()
