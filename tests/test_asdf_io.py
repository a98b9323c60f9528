"""`.asdf` reader/writer against the format of SdfGen/dllmain.cpp:250-292 and
the committed fixture."""
import os
import struct

import numpy as np
import pytest

from conftest import GOLDEN


def test_golden_file_layout_and_roundtrip(sb, tmp_path):
    path = os.path.join(GOLDEN, "sphere_d4.asdf")
    raw = open(path, "rb").read()
    assert raw[:4] == b"asdf"
    (n,) = struct.unpack("<I", raw[4:8])
    assert len(raw) == 8 + 16 * n                      # no version field (dllmain.cpp:285-286)
    od = sb.OctData.LoadAsdf(path)
    assert od.Length == n
    assert od.Structs[0, 0] == -1                       # root: Parent -1
    assert np.frombuffer(raw[8:8 + 8 * n], dtype="<i4").reshape(n, 2).tolist() == od.Structs.tolist()
    assert np.frombuffer(raw[8 + 8 * n:], dtype=np.uint8).reshape(n, 8).tolist() == od.Values.tolist()
    out = tmp_path / "copy.asdf"
    od.Save(str(out))
    assert open(out, "rb").read() == raw


def test_generator_reproduces_the_committed_fixture(sb):
    od = sb.OctData.LoadAsdf(os.path.join(GOLDEN, "sphere_d4.asdf"))
    gen = sb.sphere_d4()
    assert (gen.Structs == od.Structs).all() and (gen.Values == od.Values).all()


def test_truncated_and_foreign_files_are_io_errors(sb, tmp_path):
    raw = open(os.path.join(GOLDEN, "sphere_d4.asdf"), "rb").read()
    bad = tmp_path / "short.asdf"
    bad.write_bytes(raw[: len(raw) // 2])
    with pytest.raises(sb.SdfHipError) as e:
        sb.OctData.LoadAsdf(str(bad))
    assert e.value.code == sb._lib.ERR_IO
    bad.write_bytes(b"ply\n" + raw[4:])
    with pytest.raises(sb.SdfHipError):
        sb.OctData.LoadAsdf(str(bad))
    bad.write_bytes(b"asdf" + struct.pack("<I", 0))
    with pytest.raises(sb.SdfHipError):
        sb.OctData.LoadAsdf(str(bad))
    bad.write_bytes(b"")
    with pytest.raises(sb.SdfHipError):
        sb.OctData.LoadAsdf(str(bad))
