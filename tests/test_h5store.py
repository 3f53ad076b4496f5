"""CPU: the HDF5-layout feature store (deepmerge_amd/h5store.py; upstream: FeatureIO.save_h5 / ReadFeatures / GetFeaturesByID,
ExtractFeatures.py:88-117).  h5py is absent, so the bytes are checked against the HDF5 File Format Specification field by field
(structural validation) and round-tripped through the independent reader."""
import os
import struct

import numpy as np
import pytest

from deepmerge_amd.h5store import H5FeatureReader, H5FeatureWriter, UNDEF


def _write(path, batches, **kw):
    with H5FeatureWriter(path, **kw) as w:
        for b in batches:
            w.append(b)


def test_round_trip_appends_like_save_h5(tmp_path):
    rng = np.random.default_rng(0)
    batches = [rng.normal(size=(n, 100)).astype(np.float32) for n in (2000, 2000, 1643)]       # batch_size 2000 + a ragged tail
    path = str(tmp_path / "features.h5")
    _write(path, batches)
    want = np.concatenate(batches)
    r = H5FeatureReader(path)
    assert r.shape == want.shape and r.maxshape == (None, 100) and len(r) == want.shape[0]
    assert np.array_equal(r.rows(0, want.shape[0]), want)
    for i in (0, 1023, 1024, 1999, 2000, 5642, -1):
        assert np.array_equal(r[i], want[i])                  # GetFeaturesByID
    with pytest.raises(IndexError):
        r.rows(0, want.shape[0] + 1)
    assert r.info["eof"] == os.path.getsize(path)
    r.close()


def test_superblock_and_headers_follow_the_specification(tmp_path):
    path = str(tmp_path / "f.h5")
    _write(path, [np.arange(300 * 100, dtype=np.float32).reshape(300, 100)], chunk_rows=128)
    raw = open(path, "rb").read()
    assert raw[:8] == b"\x89HDF\r\n\x1a\n"
    assert raw[8:16] == bytes([0, 0, 0, 0, 0, 8, 8, 0])      # versions 0, 8-byte offsets and lengths
    leaf_k, int_k, flags = struct.unpack("<HHI", raw[16:24])
    assert (leaf_k, int_k, flags) == (4, 16, 0)
    base, free, eof, drv = struct.unpack("<QQQQ", raw[24:56])
    assert base == 0 and free == UNDEF and drv == UNDEF and eof == len(raw)
    root_hdr = struct.unpack("<Q", raw[64:72])[0]
    assert raw[root_hdr] == 1 and struct.unpack("<H", raw[root_hdr + 2:root_hdr + 4])[0] == 1      # v1 header, one message
    assert struct.unpack("<H", raw[root_hdr + 16:root_hdr + 18])[0] == 0x0011                      # ... the symbol table message
    r = H5FeatureReader(path)
    hdr = r.info["dataset_header"]
    assert hdr % 8 == 0 and raw[hdr] == 1
    nmsg = struct.unpack("<H", raw[hdr + 2:hdr + 4])[0]
    types, off = [], hdr + 16
    for _ in range(nmsg):
        t, size = struct.unpack("<HH", raw[off:off + 4])
        assert size % 8 == 0
        types.append(t); off += 8 + size
    assert types == [0x0001, 0x0003, 0x0005, 0x0008]          # dataspace, datatype, fill value, layout
    assert r.info["chunk_dims"] == (128, 100, 4) and r.chunk_rows == 128
    assert sorted(r.chunks) == [(0, 0), (128, 0), (256, 0)]  # 300 rows -> 3 chunks, the last stored whole
    idx = r.info["index"]
    assert raw[idx:idx + 4] == b"TREE" and raw[idx + 4] == 1 and raw[idx + 5] == 0 and struct.unpack("<H", raw[idx + 6:idx + 8])[0] == 3
    assert struct.unpack("<QQ", raw[idx + 8:idx + 24]) == (UNDEF, UNDEF)
    last_key = raw[idx + 24 + 3 * 40:idx + 24 + 3 * 40 + 32]
    assert struct.unpack("<QQQ", last_key[8:]) == (384, 0, 0)
    for (row0, _col0), addr in r.chunks.items():
        got = np.frombuffer(raw[addr:addr + 128 * 400], "<f4").reshape(128, 100)
        n = min(128, 300 - row0)
        assert np.array_equal(got[:n], np.arange(300 * 100, dtype=np.float32).reshape(300, 100)[row0:row0 + n])
    r.close()


def test_two_level_chunk_index_and_edge_sizes(tmp_path):
    rng = np.random.default_rng(1)
    data = rng.normal(size=(700, 8)).astype(np.float32)
    path = str(tmp_path / "deep.h5")
    _write(path, [data[:1], data[1:350], data[350:]], width=8, chunk_rows=4)       # 175 chunks > 64 -> leaf nodes + one level-1 node
    r = H5FeatureReader(path)
    raw = open(path, "rb").read()
    idx = r.info["index"]
    assert raw[idx + 5] == 1 and struct.unpack("<H", raw[idx + 6:idx + 8])[0] == 3         # ceil(175 / 64) leaves
    assert np.array_equal(r.rows(0, 700), data) and len(r.chunks) == 175
    r.close()
    # exactly full chunks, and an empty store
    path2 = str(tmp_path / "full.h5")
    _write(path2, [data[:8]], width=8, chunk_rows=4)
    r2 = H5FeatureReader(path2)
    assert r2.shape == (8, 8) and np.array_equal(r2.rows(0, 8), data[:8]) and os.path.getsize(path2) == r2.info["eof"]
    r2.close()
    path3 = str(tmp_path / "empty.h5")
    _write(path3, [], width=100)
    r3 = H5FeatureReader(path3)
    assert r3.shape == (0, 100) and r3.info["index"] == UNDEF
    r3.close()
    with pytest.raises(ValueError):
        H5FeatureWriter(str(tmp_path / "x.h5"), width=100).append(np.zeros((3, 99), np.float32))
    with pytest.raises(KeyError):
        H5FeatureReader(path2, name="other")


def test_featureio_store_round_trip_without_a_gpu(tmp_path):
    """FeatureIO.save_h5 / ReadFeatures / GetFeaturesByID on a CPU tensor (the class only needs the GPU for the encoder)."""
    import torch
    from deepmerge_amd.ExtractFeatures import FeatureIO
    fio = object.__new__(FeatureIO)
    fio.features = torch.randn(4100, 100)
    path = str(tmp_path / "f.h5")
    assert fio.save_h5(path, batch_size=2000) == 4100
    fio.features = None
    fio.ReadFeatures(path)
    r = H5FeatureReader(path)
    assert r.shape == (4100, 100)
    row = fio.GetFeaturesByID(4099)
    assert row.dtype == np.float32 and np.array_equal(row, r[4099])
    with pytest.raises(IndexError):
        fio.GetFeaturesByID(4100)
    fio.Close(); r.close()


def test_reader_takes_h5py_style_layouts(tmp_path):
    """What h5py's `chunks=True` / libhdf5 produce for the reference's `create_dataset("dataset", ..., maxshape=(None, 100), chunks=True)`
    (ExtractFeatures.py:88-101) and this writer's default does not: a 2-D chunk GRID (both dimensions split, e.g. (128, 25); edge
    chunks hang over the width and are stored whole), an object header whose layout message sits in a continuation block, NIL padding
    and a message type the reader does not need.  No h5py here: the vectors come from the writer's own options, the
    reader takes every size and address from the file."""
    rng = np.random.default_rng(5)
    data = rng.normal(size=(300, 100)).astype(np.float32)
    for kw in ({"chunk_cols": 25}, {"chunk_cols": 32}, {"chunk_cols": 25, "split_header": True}, {"split_header": True}):
        path = str(tmp_path / "grid.h5")
        with H5FeatureWriter(path, width=100, chunk_rows=128, **kw) as w:
            w.append(data[:70]); w.append(data[70:])
        r = H5FeatureReader(path)
        cc = kw.get("chunk_cols", 100)
        assert r.shape == (300, 100) and r.info["chunk_dims"] == (128, cc, 4) and (r.chunk_rows, r.chunk_cols) == (128, cc)
        assert len(r.chunks) == 3 * -(-100 // cc)
        assert np.array_equal(r.rows(0, 300), data) and np.array_equal(r[299], data[299]) and np.array_equal(r.rows(120, 140), data[120:140])
        raw = open(path, "rb").read()
        assert os.path.getsize(path) == r.info["eof"]
        hdr = r.info["dataset_header"]
        nmsg, first = struct.unpack("<H", raw[hdr + 2:hdr + 4])[0], struct.unpack("<I", raw[hdr + 8:hdr + 12])[0]
        types, off = [], hdr + 16
        while off < hdr + 16 + first:
            t, size = struct.unpack("<HH", raw[off:off + 4])
            types.append(t); off += 8 + size
        if kw.get("split_header"):
            assert types == [0x0001, 0x0003, 0x0005, 0x0000, 0x0010] and nmsg == 7            # + layout and modification time in the block
            caddr, clen = struct.unpack("<QQ", raw[off - 16:off])
            assert struct.unpack("<H", raw[caddr:caddr + 2])[0] == 0x0008 and struct.unpack("<H", raw[caddr + clen - 16:caddr + clen - 14])[0] == 0x0012
        else:
            assert types == [0x0001, 0x0003, 0x0005, 0x0008] and nmsg == 4
        # chunk keys are in row-major grid order and every chunk is stored whole
        keys = sorted(r.chunks)
        assert keys == [(r0, c0) for r0 in (0, 128, 256) for c0 in range(0, 100, cc)]
        (r0, c0), addr = keys[-1], r.chunks[keys[-1]]
        got = np.frombuffer(raw[addr:addr + 128 * cc * 4], "<f4").reshape(128, cc)
        assert np.array_equal(got[:300 - r0, :100 - c0], data[r0:, c0:]) and not got[300 - r0:].any()
        r.close()
    with pytest.raises(ValueError):
        H5FeatureWriter(str(tmp_path / "bad.h5"), width=100, chunk_cols=101)

