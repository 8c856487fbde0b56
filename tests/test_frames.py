"""Codec-free frame sources (visual_slam_amd/frames.py): Y4M and raw gray files, CPU only."""
import numpy as np
import pytest

from visual_slam_amd import frames, synth

H, W, N = 48, 64, 5


def _frames():
    return [synth.synth_frame(H, W, i, cell=12) for i in range(N)]


@pytest.mark.parametrize("cs", ["mono", "420jpeg"])
def test_y4m_round_trip(tmp_path, cs):
    src = _frames()
    p = str(tmp_path / "v.y4m")
    assert frames.write_y4m(p, src, fps=25, colourspace=cs) == N
    with frames.FrameFile(p) as f:
        assert (f.H, f.W, f.fps, f.colourspace) == (H, W, 25.0, cs) and f.isOpened()
        got = list(f)
    assert len(got) == N and all(np.array_equal(a, b) for a, b in zip(got, src))


def test_video_capture_shape_and_end_of_stream(tmp_path):
    src = _frames()
    p = str(tmp_path / "v.y4m")
    frames.write_y4m(p, src)
    cap = frames.VideoCapture(p)
    for i in range(N):
        ok, fr = cap.read()
        assert ok and fr.shape == (H, W, 3) and fr.dtype == np.uint8
        assert all(np.array_equal(fr[:, :, c], src[i]) for c in range(3))
    assert cap.read() == (False, None) and cap.read() == (False, None)
    cap.release()
    assert not cap.isOpened()


def test_raw_gray_and_truncated_tail(tmp_path):
    src = _frames()
    p = str(tmp_path / "v.gray")
    frames.write_raw_gray(p, src)
    with open(p, "ab") as f:
        f.write(b"\x01" * 100)                        # a partial frame at the end is dropped
    got = list(frames.FrameFile(p, H=H, W=W))
    assert len(got) == N and all(np.array_equal(a, b) for a, b in zip(got, src))
    with pytest.raises(ValueError):
        frames.FrameFile(p)                           # a raw file needs its size


def test_read_into_and_bad_inputs(tmp_path):
    src = _frames()
    p = str(tmp_path / "v.y4m")
    frames.write_y4m(p, src, colourspace="420jpeg")
    f = frames.FrameFile(p)
    buf = np.zeros((H, W), np.uint8)
    assert f.read_into(buf) and np.array_equal(buf, src[0]) and f.index == 1
    with pytest.raises(ValueError):
        f.read_into(np.zeros((H, W + 1), np.uint8))
    with pytest.raises(ValueError):
        frames.FrameFile(p, H=H + 1, W=W)
    bad = str(tmp_path / "bad.y4m")
    open(bad, "wb").write(b"YUV4MPEG2 W64 H48 C420p10\nFRAME\n")
    with pytest.raises(ValueError):
        frames.FrameFile(bad)
    nosize = str(tmp_path / "nosize.y4m")
    open(nosize, "wb").write(b"YUV4MPEG2 F30:1\n")
    with pytest.raises(ValueError):
        frames.FrameFile(nosize)
