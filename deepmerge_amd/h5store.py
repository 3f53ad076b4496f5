"""HDF5-layout feature store: the on-disk format `FeatureIO.save_h5` produces upstream (ExtractFeatures.py:88-101:
`h5f.create_dataset("dataset", data=..., maxshape=(None, 100), chunks=True)` then `dataset.resize` + append per batch, read back
with `f["dataset"][idx]`), written and read WITHOUT h5py (absent in this image) straight from the HDF5 File Format
Specification (version 0 superblock, version 1 object headers, symbol-table root group, version 1 B-tree chunk index --
the "earliest" layout h5py / libhdf5 emit by default for such a file and the one every libhdf5 can read).

    H5FeatureWriter(path, width=100)      .append(rows [n, width] float32) ... .close()
    H5FeatureReader(path)                 .shape, [i] -> row, .rows(a, b) -> [b-a, width]

One 2-D little-endian IEEE float32 dataset named "dataset", dimension 0 unlimited (max dims (UNLIMITED, width)), chunked
(chunk_rows x width by default), no filters, fill value undefined / incremental allocation.  The READER takes any 2-D chunk grid
(chunk_rows x chunk_cols with chunk_cols <= width -- what h5py's `chunks=True` guess produces for [P, 100] float32: both dimensions
halved until the chunk is small, e.g. (128, 25)), every chunk stored whole as the format requires, follows object-header continuation
blocks (message 0x0010) and skips NIL / unknown messages; the writer can produce such files too (`chunk_cols=`, `split_header=True`),
which is what the reader's tests are built from (no h5py-written fixture can be produced here).  The chunk index is a single leaf node for up
to 64 chunks and a two-level tree above (default indexed-storage K = 32).  The file is rewritten index-last on close(), so an
interrupted run leaves no half-valid index.

Parity status (round 4): h5py is absent, but the image carries libhdf5 1.10.6 with its tools (/opt/conda/bin/h5dump, h5repack,
/opt/conda/lib/libhdf5.so.103), and tests/test_h5store.py uses them in both directions: h5dump reads both writer layouts and dumps
the payload byte for byte; files re-written by h5repack (as is, and re-chunked to 128 x 25) and a file produced by the upstream call
sequence through libhdf5's C API (H5Dcreate2 with maxshape (UNLIMITED, 100) + H5Dset_extent + hyperslab writes = what h5py does)
read back bit-equal through H5FeatureReader.  The field-by-field walk of the specification stays (H5FeatureReader shares no layout
constants with the writer: it takes every size and address from the file).
"""
from __future__ import annotations

import struct
from typing import List, Tuple

import numpy as np

UNDEF = 0xFFFFFFFFFFFFFFFF
SIGNATURE = b"\x89HDF\r\n\x1a\n"
GROUP_LEAF_K, GROUP_INTERNAL_K, ISTORE_K = 4, 16, 32


def _pad8(b: bytes) -> bytes:
    return b + b"\0" * (-len(b) % 8)


def _message(mtype: int, data: bytes, flags: int = 0) -> bytes:
    data = _pad8(data)
    return struct.pack("<HHB3x", mtype, len(data), flags) + data


def _object_header(messages: List[bytes]) -> bytes:
    body = b"".join(messages)
    return struct.pack("<BBHII4x", 1, 0, len(messages), 1, len(body)) + body


class H5FeatureWriter:
    def __init__(self, path: str, width: int = 100, chunk_rows: int = 1024, name: str = "dataset", chunk_cols: int = 0,
                 split_header: bool = False):
        """chunk_cols: columns per chunk (0 = the full width, one chunk per row band).  split_header: put the layout message into an
        object-header continuation block behind the data (with a NIL message in front of the continuation message and a
        modification-time message beside the layout), the way libhdf5 lays a header out once it has outgrown its first block."""
        if width <= 0 or chunk_rows <= 0:
            raise ValueError("width and chunk_rows must be positive")
        self.path, self.width, self.chunk_rows, self.name = path, int(width), int(chunk_rows), name
        self.chunk_cols = int(chunk_cols) if chunk_cols else int(width)
        if not 0 < self.chunk_cols <= self.width:
            raise ValueError("chunk_cols must be in 1..width")
        self.col_chunks = (self.width + self.chunk_cols - 1) // self.chunk_cols
        self.split_header = bool(split_header)
        self.f = open(path, "wb")
        self.rows = 0
        self._tail = np.zeros((0, self.width), np.float32)      # rows of the last, partially filled chunk
        self._chunks: List[int] = []                            # file address of every full chunk written so far
        # ---- fixed metadata block; addresses are known up front, the dataset header and the index are patched in close() ----
        self.addr_root = 96
        root = _object_header([_message(0x0011, struct.pack("<QQ", 0, 0))])        # patched below once addresses are known
        self.addr_btree = self.addr_root + len(root)
        self.group_btree_size = 24 + (2 * GROUP_INTERNAL_K + 1) * 8 + 2 * GROUP_INTERNAL_K * 8
        self.addr_heap = self.addr_btree + self.group_btree_size
        self.heap_data_size = 88
        self.addr_heap_data = self.addr_heap + 32
        self.addr_snod = self.addr_heap_data + self.heap_data_size
        self.snod_size = 8 + 2 * GROUP_LEAF_K * 40
        self.addr_dset = self.addr_snod + self.snod_size
        self.dset_header_size = len(self._dataset_header(0, 0))
        self.addr_data = (self.addr_dset + self.dset_header_size + 7) // 8 * 8
        self.f.write(b"\0" * self.addr_data)
        self.chunk_bytes = self.chunk_rows * self.chunk_cols * 4

    # ---- metadata pieces -----------------------------------------------------------------------------------------------
    def _layout_message(self, index_addr: int) -> bytes:
        layout = struct.pack("<BBB", 3, 2, 3) + struct.pack("<Q", index_addr) + struct.pack("<III", self.chunk_rows, self.chunk_cols, 4)
        return _message(0x0008, layout)

    def _continuation_block(self, index_addr: int) -> bytes:
        mtime = _message(0x0012, struct.pack("<B3xI", 1, 0))        # modification time, version 1, seconds since the epoch
        return self._layout_message(index_addr) + mtime

    def _dataset_header(self, rows: int, index_addr: int, cont_addr: int = 0) -> bytes:
        dataspace = struct.pack("<BBB5x", 1, 2, 1) + struct.pack("<QQ", rows, self.width) + struct.pack("<QQ", UNDEF, self.width)
        # IEEE binary32, little-endian: class 1 (floating point), version 1; bit field: LE, no padding, mantissa normalisation 2
        # (implied leading one), sign at bit 31; properties: bit offset 0, precision 32, exponent at 23 (8 bits), mantissa at 0
        # (23 bits), bias 127
        datatype = struct.pack("<BBBBI", 0x11, 0x20, 0x1F, 0x00, 4) + struct.pack("<HHBBBBI", 0, 32, 23, 8, 0, 23, 127)
        # version 2, incremental allocation, fill written "if set", DEFAULT fill value (defined, size 0): the bytes libhdf5 1.10 /
        # h5py write for `create_dataset(..., chunks=True)` (checked against a libhdf5-written file, tests/test_h5store.py).  Rounds
        # 2-3 wrote "on allocation + undefined value", which libhdf5 READS but refuses to re-create (h5repack -l failed with
        # "fill value writing on allocation set, but no fill value defined")
        fill = struct.pack("<BBBBI", 2, 3, 2, 1, 0)
        head = [_message(0x0001, dataspace), _message(0x0003, datatype, 1), _message(0x0005, fill)]
        if not self.split_header:
            return _object_header(head + [self._layout_message(index_addr)])
        cont = _message(0x0010, struct.pack("<QQ", cont_addr, len(self._continuation_block(0))))
        body = b"".join(head + [_message(0x0000, b"\0" * 8), cont])
        # the message count of a version-1 header covers the continuation blocks too (+ layout, + modification time)
        return struct.pack("<BBHII4x", 1, 0, len(head) + 2 + 2, 1, len(body)) + body

    def _chunk_key(self, i: int, size: int) -> bytes:
        """Key of chunk number i in row-major order of the chunk grid: byte size, filter mask, element offsets (row, column, 0)."""
        return struct.pack("<II", size, 0) + struct.pack("<QQQ", (i // self.col_chunks) * self.chunk_rows, (i % self.col_chunks) * self.chunk_cols, 0)

    def _chunk_node(self, level: int, entries: List[Tuple[int, int]], last_row_chunk: int) -> bytes:
        """entries: (first chunk index, child address).  A node is allocated at full size (2K entries) as libhdf5 does."""
        body = b"TREE" + struct.pack("<BBH", 1, level, len(entries)) + struct.pack("<QQ", UNDEF, UNDEF)
        for ci, addr in entries:
            body += self._chunk_key(ci, self.chunk_bytes) + struct.pack("<Q", addr)
        body += self._chunk_key(last_row_chunk, 0)              # final key: the offset just past the last chunk of this node
        full = 24 + (2 * ISTORE_K + 1) * 32 + 2 * ISTORE_K * 8
        return body + b"\0" * (full - len(body))

    # ---- data ------------------------------------------------------------------------------------------------------------
    def append(self, rows) -> None:
        a = np.ascontiguousarray(np.asarray(rows, dtype=np.float32))
        if a.ndim != 2 or a.shape[1] != self.width:
            raise ValueError(f"rows must be [n, {self.width}], got {a.shape}")
        self.rows += a.shape[0]
        buf = np.concatenate([self._tail, a]) if self._tail.shape[0] else a
        n_full = buf.shape[0] // self.chunk_rows
        for c in range(n_full):
            self._write_band(buf[c * self.chunk_rows:(c + 1) * self.chunk_rows])
        self._tail = buf[n_full * self.chunk_rows:].copy()

    def _write_band(self, band: np.ndarray) -> None:
        """One band of chunk_rows rows as its column chunks, each stored whole (edge chunks padded, as the format requires)."""
        for cc in range(self.col_chunks):
            piece = np.zeros((self.chunk_rows, self.chunk_cols), "<f4")
            cols = band[:, cc * self.chunk_cols:(cc + 1) * self.chunk_cols]
            piece[:cols.shape[0], :cols.shape[1]] = cols
            self._chunks.append(self.f.tell())
            self.f.write(piece.tobytes())

    def close(self) -> None:
        if self.f is None:
            return
        f = self.f
        chunks = list(self._chunks)
        if self._tail.shape[0]:                                   # the last band is stored whole; rows past `rows` are never read
            self._write_band(self._tail)
            chunks = list(self._chunks)
        cont_addr = 0
        if self.split_header:
            cont_addr = f.tell()
            f.write(b"\0" * len(self._continuation_block(0)))      # patched once the index address is known
        n = len(chunks)
        index_addr = UNDEF
        if n:
            per = 2 * ISTORE_K
            if n <= per:
                index_addr = f.tell()
                f.write(self._chunk_node(0, list(enumerate(chunks)), n))
            else:
                if n > per * per:
                    raise ValueError("more than 4096 chunks: raise chunk_rows")
                leaves = []
                for s in range(0, n, per):
                    leaves.append((s, f.tell()))
                    f.write(self._chunk_node(0, [(s + i, a) for i, a in enumerate(chunks[s:s + per])], min(n, s + per)))
                index_addr = f.tell()
                f.write(self._chunk_node(1, leaves, n))
        eof = f.tell()
        # ---- metadata ----------------------------------------------------------------------------------------------------
        f.seek(0)
        sb = SIGNATURE + struct.pack("<BBBBBBBB", 0, 0, 0, 0, 0, 8, 8, 0) + struct.pack("<HHI", GROUP_LEAF_K, GROUP_INTERNAL_K, 0)
        sb += struct.pack("<QQQQ", 0, UNDEF, eof, UNDEF)
        sb += struct.pack("<QQII", 0, self.addr_root, 1, 0) + struct.pack("<QQ", self.addr_btree, self.addr_heap)      # root symbol table entry
        assert len(sb) == 96
        f.write(sb)
        f.write(_object_header([_message(0x0011, struct.pack("<QQ", self.addr_btree, self.addr_heap))]))
        name_off = 8
        node = b"TREE" + struct.pack("<BBH", 0, 0, 1) + struct.pack("<QQ", UNDEF, UNDEF) + struct.pack("<QQQ", 0, self.addr_snod, name_off)
        f.write(node + b"\0" * (self.group_btree_size - len(node)))
        name = self.name.encode() + b"\0"
        seg = b"\0" * 8 + _pad8(name)                              # offset 0: the empty name of the root; offset 8: the dataset's
        free_off = len(seg)
        if free_off + 16 > self.heap_data_size:
            raise ValueError("dataset name too long for the heap segment")
        seg += struct.pack("<QQ", 1, self.heap_data_size - free_off)      # one free block: next = 1 (none), its size
        f.write(b"HEAP" + struct.pack("<B3x", 0) + struct.pack("<QQQ", self.heap_data_size, free_off, self.addr_heap_data))
        f.write(seg + b"\0" * (self.heap_data_size - len(seg)))
        f.seek(self.addr_snod)
        snod = b"SNOD" + struct.pack("<BBH", 1, 0, 1) + struct.pack("<QQII16x", name_off, self.addr_dset, 0, 0)
        f.write(snod + b"\0" * (self.snod_size - len(snod)))
        hdr = self._dataset_header(self.rows, index_addr, cont_addr)
        assert len(hdr) == self.dset_header_size
        f.write(hdr)
        if self.split_header:
            f.seek(cont_addr)
            f.write(self._continuation_block(index_addr))
        f.close()
        self.f = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


class H5FeatureReader:
    """Walks the file the way the specification describes it: superblock -> root symbol-table entry -> group B-tree / local heap
    -> symbol node -> dataset object header (dataspace, datatype, layout) -> chunk B-tree -> raw chunks."""

    def __init__(self, path: str, name: str = "dataset"):
        self.f = open(path, "rb")
        self.info = {}
        self._parse(name)

    def _at(self, addr: int, n: int) -> bytes:
        self.f.seek(addr)
        b = self.f.read(n)
        if len(b) != n:
            raise ValueError(f"truncated file: {n} bytes at {addr}")
        return b

    def _parse(self, name: str) -> None:
        sb = self._at(0, 96)
        if sb[:8] != SIGNATURE:
            raise ValueError("not an HDF5 file")
        ver, _fs, _rg, _r, _sh, so, sl = struct.unpack("<BBBBBBB", sb[8:15])
        if ver != 0 or so != 8 or sl != 8:
            raise ValueError("this reader handles version-0 superblocks with 8-byte offsets / lengths")
        leaf_k, int_k = struct.unpack("<HH", sb[16:20])
        base, _free, eof, _drv = struct.unpack("<QQQQ", sb[24:56])
        _lno, root_hdr, cache, _res = struct.unpack("<QQII", sb[56:80])
        btree, heap = struct.unpack("<QQ", sb[80:96])
        self.info.update(superblock_version=ver, group_leaf_k=leaf_k, group_internal_k=int_k, eof=eof, root_header=root_hdr)
        if cache != 1:                                       # not cached: take B-tree / heap from the root header's symbol-table message
            btree, heap = self._symbol_table_message(root_hdr)
        elif self._symbol_table_message(root_hdr) != (btree, heap):
            raise ValueError("root symbol-table entry cache disagrees with the root object header")
        # local heap
        hp = self._at(heap, 32)
        if hp[:4] != b"HEAP":
            raise ValueError("bad local heap signature")
        seg_size, free_head, seg_addr = struct.unpack("<QQQ", hp[8:32])
        heap_data = self._at(seg_addr, seg_size)
        self.info.update(heap_segment=seg_size, heap_free_head=free_head)

        def heap_str(off):
            end = heap_data.index(b"\0", off)
            return heap_data[off:end].decode()
        # group B-tree (leaf level only is needed for one dataset, but follow levels generically)
        target = None
        stack = [btree]
        while stack:
            node = stack.pop()
            hd = self._at(node, 24)
            if hd[:4] != b"TREE" or hd[4] != 0:
                raise ValueError("bad group B-tree node")
            level, used = hd[5], struct.unpack("<H", hd[6:8])[0]
            body = self._at(node + 24, (2 * used + 1) * 8)
            children = [struct.unpack("<Q", body[8 + 16 * i:16 + 16 * i])[0] for i in range(used)]
            if level > 0:
                stack.extend(children)
                continue
            for sn in children:
                sh = self._at(sn, 8)
                if sh[:4] != b"SNOD":
                    raise ValueError("bad symbol table node")
                nsym = struct.unpack("<H", sh[6:8])[0]
                for i in range(nsym):
                    e = self._at(sn + 8 + 40 * i, 40)
                    noff, ohdr = struct.unpack("<QQ", e[:16])
                    if heap_str(noff) == name:
                        target = ohdr
        if target is None:
            raise KeyError(name)
        self.info["dataset_header"] = target
        msgs = self._messages(target)
        sp = msgs[0x0001]
        sver, rank, flags = sp[0], sp[1], sp[2]
        if sver != 1 or rank != 2:
            raise ValueError("expected a version-1 rank-2 dataspace")
        dims = struct.unpack("<QQ", sp[8:24])
        maxdims = struct.unpack("<QQ", sp[24:40]) if flags & 1 else dims
        dt = msgs[0x0003]
        cls, ver_dt = dt[0] & 0x0F, dt[0] >> 4
        size = struct.unpack("<I", dt[4:8])[0]
        boff, prec, eloc, esz, mloc, msz, bias = struct.unpack("<HHBBBBI", dt[8:20])
        if not (cls == 1 and size == 4 and (dt[1] & 1) == 0 and prec == 32 and eloc == 23 and esz == 8 and mloc == 0 and msz == 23 and bias == 127 and dt[2] == 31):
            raise ValueError("dataset is not little-endian IEEE float32")
        lay = msgs[0x0008]
        if lay[0] != 3 or lay[1] != 2:
            raise ValueError("expected a version-3 chunked layout")
        ndim = lay[2]
        index = struct.unpack("<Q", lay[3:11])[0]
        cdims = struct.unpack("<" + "I" * ndim, lay[11:11 + 4 * ndim])
        if ndim != 3 or cdims[2] != 4 or not (0 < cdims[1] <= dims[1]) or cdims[0] == 0:
            raise ValueError(f"unsupported chunk shape {cdims}")
        self.shape = (int(dims[0]), int(dims[1]))
        self.maxshape = (None if maxdims[0] == UNDEF else int(maxdims[0]), int(maxdims[1]))
        self.chunk_rows, self.chunk_cols = int(cdims[0]), int(cdims[1])
        self.info.update(datatype_version=ver_dt, fill=tuple(msgs[0x0005][:4]) if 0x0005 in msgs else None, chunk_dims=cdims, index=index)
        # chunk index
        self.chunks = {}
        if index != UNDEF:
            self._walk_chunks(index)
        need = (self.shape[0] + self.chunk_rows - 1) // self.chunk_rows
        ncol = (self.shape[1] + self.chunk_cols - 1) // self.chunk_cols
        missing = [(r, c) for r in range(need) for c in range(ncol) if (r * self.chunk_rows, c * self.chunk_cols) not in self.chunks]
        if missing:
            raise ValueError(f"chunks missing from the index: {missing[:5]}")

    def _symbol_table_message(self, hdr: int):
        m = self._messages(hdr)
        return struct.unpack("<QQ", m[0x0011][:16])

    def _messages(self, addr: int) -> dict:
        """Messages of a version-1 object header by type, continuation blocks (0x0010: address, length) included; the header's message
        count covers all blocks.  NIL messages (0x0000) are padding."""
        ver, _r, nmsg, _ref, size = struct.unpack("<BBHII", self._at(addr, 12))
        if ver != 1:
            raise ValueError("expected a version-1 object header")
        blocks = [(addr + 16, size)]
        out, seen = {}, 0
        while blocks and seen < nmsg:
            baddr, bsize = blocks.pop(0)
            body = self._at(baddr, bsize)
            off = 0
            while off + 8 <= len(body) and seen < nmsg:
                mtype, msize, _flags = struct.unpack("<HHB", body[off:off + 5])
                data = body[off + 8:off + 8 + msize]
                if len(data) != msize:
                    raise ValueError("object header message runs past its block")
                if mtype == 0x0010:
                    blocks.append(struct.unpack("<QQ", data[:16]))
                elif mtype != 0x0000:
                    out[mtype] = data
                seen += 1
                off += 8 + msize
        if seen != nmsg:
            raise ValueError(f"object header announces {nmsg} messages, found {seen}")
        return out

    def _walk_chunks(self, node: int) -> None:
        hd = self._at(node, 24)
        if hd[:4] != b"TREE" or hd[4] != 1:
            raise ValueError("bad chunk B-tree node")
        level, used = hd[5], struct.unpack("<H", hd[6:8])[0]
        body = self._at(node + 24, used * 40 + 32)
        for i in range(used):
            key = body[40 * i:40 * i + 32]
            size, mask = struct.unpack("<II", key[:8])
            r0, c0, _e = struct.unpack("<QQQ", key[8:32])
            child = struct.unpack("<Q", body[40 * i + 32:40 * i + 40])[0]
            if level > 0:
                self._walk_chunks(child)
            else:
                if mask != 0 or size != self.chunk_rows * self.chunk_cols * 4 or r0 % self.chunk_rows or c0 % self.chunk_cols:
                    raise ValueError("filtered / partial / misaligned chunks are not supported")
                self.chunks[(int(r0), int(c0))] = child

    def rows(self, a: int, b: int) -> np.ndarray:
        if not (0 <= a <= b <= self.shape[0]):
            raise IndexError("index error!")
        out = np.empty((b - a, self.shape[1]), np.float32)
        r = a
        while r < b:
            r0 = r // self.chunk_rows * self.chunk_rows
            n = min(b, r0 + self.chunk_rows) - r
            for c0 in range(0, self.shape[1], self.chunk_cols):      # the band's column chunks; the last one may hang over the width
                w = min(self.chunk_cols, self.shape[1] - c0)
                raw = self._at(self.chunks[(r0, c0)] + (r - r0) * self.chunk_cols * 4, n * self.chunk_cols * 4)
                out[r - a:r - a + n, c0:c0 + w] = np.frombuffer(raw, "<f4").reshape(n, self.chunk_cols)[:, :w]
            r += n
        return out

    def __getitem__(self, i: int) -> np.ndarray:
        i = int(i)
        if i < 0:
            i += self.shape[0]
        return self.rows(i, i + 1)[0]

    def __len__(self):
        return self.shape[0]

    def close(self):
        self.f.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
