"""CPU: the HDF5-layout feature store (deepmerge_amd/h5store.py; upstream: FeatureIO.save_h5 / ReadFeatures / GetFeaturesByID,
ExtractFeatures.py:88-117).  h5py is absent; the bytes are checked against the HDF5 File Format Specification field by field,
round-tripped through the independent reader, and -- where the image carries libhdf5 (/opt/conda: h5dump, h5repack,
libhdf5.so.103) -- validated by the library itself in both directions (round 4)."""
import ctypes as C
import os
import shutil
import struct
import subprocess

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



# ---- libhdf5 itself (round 4) ----------------------------------------------------------------------------------
def _tool(name):
    for cand in (shutil.which(name), os.path.join("/opt/conda/bin", name)):
        if cand and os.path.exists(cand):
            return cand
    return None


H5DUMP, H5REPACK = _tool("h5dump"), _tool("h5repack")
LIBHDF5 = next((p for p in ("/opt/conda/lib/libhdf5.so.103", "/opt/conda/lib/libhdf5.so") if os.path.exists(p)), None)
needs_tools = pytest.mark.skipif(H5DUMP is None or H5REPACK is None, reason="h5dump / h5repack not in this image")


def _features(rows=3000, seed=0):
    return np.random.default_rng(seed).standard_normal((rows, 100)).astype(np.float32)


@needs_tools
@pytest.mark.parametrize("kw", [{}, {"chunk_cols": 25, "split_header": True}], ids=["default", "column_chunks_split_header"])
def test_libhdf5_reads_the_writers_files(tmp_path, kw):
    """h5dump (libhdf5) opens both layouts the writer produces, reports the dataset upstream creates (ExtractFeatures.py:88-101:
    float32, (rows, 100) with an unlimited first dimension, default fill value, incremental allocation) and dumps the payload
    byte for byte."""
    x = _features()
    path, out = str(tmp_path / "w.h5"), str(tmp_path / "payload.bin")
    _write(path, [x[i:i + 700] for i in range(0, 3000, 700)], **kw)
    head = subprocess.run([H5DUMP, "-p", "-H", path], capture_output=True, text=True, check=True).stdout
    assert "H5T_IEEE_F32LE" in head and "( 3000, 100 ) / ( H5S_UNLIMITED, 100 )" in head
    assert "CHUNKED ( 1024, %d )" % kw.get("chunk_cols", 100) in head
    assert "H5D_FILL_TIME_IFSET" in head and "H5D_FILL_VALUE_DEFAULT" in head and "H5D_ALLOC_TIME_INCR" in head
    subprocess.run([H5DUMP, "-d", "/dataset", "-b", "LE", "-o", out, path], capture_output=True, text=True, check=True)
    assert np.array_equal(np.fromfile(out, dtype="<f4").reshape(-1, 100), x)


@needs_tools
def test_reader_takes_files_rewritten_by_libhdf5(tmp_path):
    """h5repack rewrites a writer file with libhdf5's own writer -- as is, and re-chunked to h5py's `chunks=True` guess for this
    dataset class (128 x 25) -- and H5FeatureReader reads both back bit-equal.  The re-chunk leg is the one that failed in round 3:
    libhdf5 refused to re-create a dataset whose fill message said "write on allocation" with an undefined value."""
    x = _features()
    src = str(tmp_path / "w.h5")
    _write(src, [x[i:i + 700] for i in range(0, 3000, 700)])
    for name, args, chunk in (("plain.h5", [], (1024, 100, 4)), ("rechunked.h5", ["-l", "dataset:CHUNK=128x25"], (128, 25, 4))):
        dst = str(tmp_path / name)
        r = subprocess.run([H5REPACK] + args + [src, dst], capture_output=True, text=True)
        assert r.returncode == 0 and "could not create" not in (r.stdout + r.stderr), r.stdout + r.stderr
        with H5FeatureReader(dst) as rd:
            assert rd.shape == (3000, 100) and rd.info["chunk_dims"] == chunk
            assert np.array_equal(rd.rows(0, 3000), x) and np.array_equal(rd[1234], x[1234])


@pytest.mark.skipif(LIBHDF5 is None, reason="libhdf5 not in this image")
def test_reader_takes_the_file_h5py_would_write(tmp_path):
    """The upstream sequence itself through libhdf5's C API (what h5py wraps): H5Dcreate2 with maxshape (UNLIMITED, 100) and
    chunks (128, 25), then H5Dset_extent + a hyperslab write per batch (`dataset.resize` + append, ExtractFeatures.py:95-101).
    H5FeatureReader reads it bit-equal, and its fill-value message equals the one H5FeatureWriter emits."""
    L = C.CDLL(LIBHDF5)
    hid, hsz = C.c_int64, C.c_uint64
    L.H5open()
    glob = lambda n: hid.in_dll(L, n).value      # noqa: E731
    L.H5Fcreate.restype = hid; L.H5Fcreate.argtypes = [C.c_char_p, C.c_uint, hid, hid]
    L.H5Screate_simple.restype = hid; L.H5Screate_simple.argtypes = [C.c_int, C.POINTER(hsz), C.POINTER(hsz)]
    L.H5Pcreate.restype = hid; L.H5Pcreate.argtypes = [hid]
    L.H5Pset_chunk.argtypes = [hid, C.c_int, C.POINTER(hsz)]
    L.H5Dcreate2.restype = hid; L.H5Dcreate2.argtypes = [hid, C.c_char_p, hid, hid, hid, hid, hid]
    L.H5Dset_extent.argtypes = [hid, C.POINTER(hsz)]
    L.H5Dget_space.restype = hid; L.H5Dget_space.argtypes = [hid]
    L.H5Sselect_hyperslab.argtypes = [hid, C.c_int, C.POINTER(hsz), C.POINTER(hsz), C.POINTER(hsz), C.POINTER(hsz)]
    L.H5Dwrite.argtypes = [hid, hid, hid, hid, hid, C.c_void_p]
    for f in ("H5Dclose", "H5Sclose", "H5Pclose", "H5Fclose"):
        getattr(L, f).argtypes = [hid]
    x = _features()
    path = str(tmp_path / "lib.h5")
    f = L.H5Fcreate(path.encode(), 2, 0, 0)                              # H5F_ACC_TRUNC
    sp = L.H5Screate_simple(2, (hsz * 2)(700, 100), (hsz * 2)(0xFFFFFFFFFFFFFFFF, 100))
    dcpl = L.H5Pcreate(glob("H5P_CLS_DATASET_CREATE_ID_g"))
    assert L.H5Pset_chunk(dcpl, 2, (hsz * 2)(128, 25)) >= 0
    d = L.H5Dcreate2(f, b"dataset", glob("H5T_IEEE_F32LE_g"), sp, 0, dcpl, 0)
    assert f >= 0 and sp >= 0 and dcpl >= 0 and d >= 0
    for i in range(0, 3000, 700):
        n = min(700, 3000 - i)
        assert L.H5Dset_extent(d, (hsz * 2)(i + n, 100)) >= 0
        fs = L.H5Dget_space(d)
        cnt = (hsz * 2)(n, 100)
        assert L.H5Sselect_hyperslab(fs, 0, (hsz * 2)(i, 0), None, cnt, None) >= 0          # H5S_SELECT_SET
        ms = L.H5Screate_simple(2, cnt, None)
        blk = np.ascontiguousarray(x[i:i + n])
        assert L.H5Dwrite(d, glob("H5T_NATIVE_FLOAT_g"), ms, fs, 0, blk.ctypes.data) >= 0
        L.H5Sclose(ms); L.H5Sclose(fs)
    L.H5Dclose(d); L.H5Sclose(sp); L.H5Pclose(dcpl); L.H5Fclose(f)
    with H5FeatureReader(path) as rd:
        assert rd.shape == (3000, 100) and rd.info["chunk_dims"] == (128, 25, 4)
        assert np.array_equal(rd.rows(0, 3000), x) and np.array_equal(rd[2999], x[2999])
        lib_fill = rd.info["fill"]
    ours = str(tmp_path / "w.h5")
    _write(ours, [x[:10]])
    with H5FeatureReader(ours) as rd:
        assert rd.info["fill"] == lib_fill == (2, 3, 2, 1)
