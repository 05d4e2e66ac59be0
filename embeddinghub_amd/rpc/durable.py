"""Durable store with rebuild-on-load for the gRPC service.

The reference keeps every embedding in RocksDB as a serialized `Embedding` proto (storage.cc:28-31,
serializer.cc:13-27: tag 0x0A, varint byte length, packed little-endian fp32), its catalog as
`SpaceEntry` / `VersionEntry` protos (embedding_store_meta.proto:9-19), and rebuilds a space's index after a
restart by replaying N sequential `ANNIndex::set` calls (version.cc:64-74) — the worst behaviour of the
reference at 10 M+ rows.  Here the same value and catalog encodings go to append-only logs, and a restart
streams them back through the engine's bulk write path (`set_batch`, ~1.3 M rows/s at d=1536) instead of
one insertion at a time.

Per space directory:  values.dat  fixed-size records, each the reference's serialized Embedding
                      keylens.u32 / keys.bin  the keys, in record order
Upserts append (last record of a key wins on replay, as sequential Sets would leave it).  A torn tail (a
crash mid-append) is cut back to the last complete record.  catalog.log: length-prefixed records
(op byte, SpaceEntry, VersionEntry).
"""
import os
import shutil
import struct
import threading

import numpy as np

from . import embedding_store_pb2 as pb
from .server import SpaceNotWritable

_CREATE, _DELETE, _FREEZE = 1, 2, 3
DEFAULT_VERSION = "initial"  # server.cc:28


def _varint(n):
    out = bytearray()
    while True:
        b = n & 0x7F
        n >>= 7
        out.append(b | (0x80 if n else 0))
        if not n:
            return bytes(out)


def value_header(dims):
    """First bytes of the serialized Embedding of a dims-long vector: field 1, wire type 2, byte length."""
    return b"\x0a" + _varint(4 * dims)


class _SpaceLog:
    def __init__(self, path, dims, sync):
        self.dims, self.sync = dims, sync
        self.head = value_header(dims)
        self.rec = len(self.head) + 4 * dims
        os.makedirs(path, exist_ok=True)
        self.paths = [os.path.join(path, n) for n in ("values.dat", "keylens.u32", "keys.bin")]
        self._repair()
        self.f = [open(p, "ab") for p in self.paths]

    def _repair(self):
        """Cut a torn tail back to the last record present in all three files."""
        sizes = [os.path.getsize(p) if os.path.exists(p) else 0 for p in self.paths]
        n = min(sizes[0] // self.rec, sizes[1] // 4)
        if n:
            lens = np.fromfile(self.paths[1], dtype="<u4", count=n)
            ends = np.cumsum(lens, dtype=np.int64)
            n = int(np.searchsorted(ends, sizes[2], side="right"))
            key_bytes = int(ends[n - 1]) if n else 0
        else:
            key_bytes = 0
        for p, size in zip(self.paths, (n * self.rec, n * 4, key_bytes)):
            if os.path.exists(p) and os.path.getsize(p) != size:
                with open(p, "r+b") as f:
                    f.truncate(size)
        self.count = n

    def append(self, keys, vecs):
        v = np.ascontiguousarray(vecs, dtype="<f4").reshape(len(keys), self.dims)
        rec = np.empty((len(keys), self.rec), dtype=np.uint8)
        rec[:, :len(self.head)] = np.frombuffer(self.head, dtype=np.uint8)
        rec[:, len(self.head):] = v.view(np.uint8)
        kb = [k.encode() if isinstance(k, str) else bytes(k) for k in keys]
        # keys first, the value record last: a record counts only once its value is complete
        self.f[2].write(b"".join(kb))
        self.f[1].write(np.array([len(k) for k in kb], dtype="<u4").tobytes())
        self.f[2].flush()
        self.f[1].flush()
        self.f[0].write(rec.tobytes())
        self.f[0].flush()
        if self.sync:
            for f in self.f:
                os.fsync(f.fileno())
        self.count += len(keys)

    def mark(self):
        """sizes of the three files: a point rollback() can return to"""
        for f in self.f:
            f.flush()
        return [os.path.getsize(p) for p in self.paths], self.count

    def rollback(self, mark):
        """cut the log back to `mark` (an append whose rows the engine then refused must not survive a restart)"""
        sizes, count = mark
        for f, size in zip(self.f, sizes):
            f.flush()
            f.truncate(size)
        self.count = count

    def replay(self, chunk=65536):
        """Yield (keys, vectors[n, dims]) chunks in append order."""
        if not self.count:
            return
        lens = np.fromfile(self.paths[1], dtype="<u4", count=self.count)
        with open(self.paths[2], "rb") as f:
            blob = f.read(int(lens.sum()))
        ends = np.cumsum(lens, dtype=np.int64)
        recs = np.memmap(self.paths[0], dtype=np.uint8, mode="r", shape=(self.count, self.rec))
        h = len(self.head)
        for i0 in range(0, self.count, chunk):
            i1 = min(self.count, i0 + chunk)
            if not (recs[i0:i1, :h] == np.frombuffer(self.head, dtype=np.uint8)).all():
                raise ValueError("values.dat: record header does not match dims=%d" % self.dims)
            vecs = np.ascontiguousarray(recs[i0:i1, h:]).view("<f4").reshape(i1 - i0, self.dims)
            start = int(ends[i0 - 1]) if i0 else 0
            offs = np.concatenate(([start], ends[i0:i1]))
            keys = [blob[offs[j]:offs[j + 1]].decode() for j in range(i1 - i0)]
            yield keys, vecs

    def close(self):
        for f in self.f:
            f.close()


class DurableSpace:
    def __init__(self, inner, log, owner, name):
        self._inner, self._log, self._owner, self._name = inner, log, owner, name
        self.dims = inner.dims
        self._mu = threading.Lock()
        self._freeze_waiting = 0   # FreezeSpace calls waiting for the space: writers stand back while it is > 0

    def set(self, key, vec):
        self.set_batch([key], [vec])

    def set_batch(self, keys, vecs):
        # A pending freeze goes first: under back-to-back writes the space's lock is free for microseconds at a time and a
        # freeze that only TRIES the lock (below) could miss it for ever.  A writer that finds a freeze waiting
        # steps aside until it is through — and is then refused by the engine ("Cannot write to immutable space"), as the
        # reference's mutex order would have it.
        import time as _time
        while self._freeze_waiting:
            _time.sleep(0.001)
        with self._mu:  # the log order is the apply order
            # log first, then apply: a write is never served without being persisted.  If the engine refuses the
            # rows (immutable space, allocation failure ...) the log is cut back, so a restart does not replay them.
            mark = self._log.mark()
            self._log.append(keys, vecs)
            try:
                self._inner.set_batch(keys, vecs)
            except BaseException:
                self._log.rollback(mark)
                raise

    def freeze(self):
        # lock order everywhere: the store's lock BEFORE a space's (delete_space holds the store's lock and then takes
        # the space's; taking them the other way round here deadlocked a concurrent FreezeSpace / DeleteSpace)
        # ... but the store's lock is never HELD while waiting for a busy space (a long set_batch on this space — log
        # append + engine write — would stall get_space / create_space / delete_space of every other space): the space's
        # lock is TRIED (never waited for) under the store's, and if the space is busy the store's lock is let go and this
        # thread sleeps before the next try — Python locks are not fair: a loop that re-took the store's lock at once kept
        # it away from everybody else for as long as the space stayed busy.
        import time as _time
        deadline = _time.monotonic() + 120.0
        self._freeze_waiting += 1      # (under the GIL; writers poll it before they take the space's lock)
        try:
            while True:
                with self._owner._mu:
                    if self._mu.acquire(blocking=False):
                        try:
                            self._inner.freeze()
                            if self._owner._spaces.get(self._name) is self:  # (a stale handle of a deleted space records nothing)
                                self._owner._catalog(_FREEZE, self._name, self.dims)
                        finally:
                            self._mu.release()
                        return
                if _time.monotonic() > deadline:   # (one write cannot take this long; a bounded wait, not a hang)
                    raise TimeoutError("FreezeSpace(%r): the space stayed busy for 120 s" % self._name)
                _time.sleep(0.002)
        finally:
            self._freeze_waiting -= 1

    def __getattr__(self, name):  # get / nearest / keys_sorted / __len__ ... : straight through
        return getattr(self._inner, name)

    def __len__(self):
        return len(self._inner)


class DurableStore:
    """Wraps any store of server.py (EngineStore in production) with logs under `data_dir`; constructing it
    on a directory that already holds logs rebuilds every space through the inner store's bulk write path."""

    def __init__(self, inner, data_dir, sync=False):
        self._inner, self._dir, self._sync = inner, data_dir, sync
        self._mu = threading.RLock()  # held across a whole create / delete: the two never interleave for one name
        self._spaces = {}
        self.rebuilt_rows = 0
        os.makedirs(data_dir, exist_ok=True)
        self._cat_path = os.path.join(data_dir, "catalog.log")
        live = self._read_catalog()
        self._cat = open(self._cat_path, "ab")
        for name, (dims, frozen) in live.items():
            sp = self._open(name, dims)
            for keys, vecs in sp._log.replay():
                sp._inner.set_batch(keys, vecs)
                self.rebuilt_rows += len(keys)
            if frozen:
                sp._inner.freeze()

    def _space_dir(self, name):
        return os.path.join(self._dir, "space-" + name.encode().hex())

    def _read_catalog(self):
        live = {}
        if not os.path.exists(self._cat_path):
            return live
        with open(self._cat_path, "rb") as f:
            data = f.read()
        pos = good = 0
        while pos + 4 <= len(data):
            (n,) = struct.unpack_from("<I", data, pos)
            if pos + 4 + n > len(data):
                break
            body = data[pos + 4:pos + 4 + n]
            pos += 4 + n
            good = pos
            op = body[0]
            (sn,) = struct.unpack_from("<I", body, 1)
            se = pb.SpaceEntry.FromString(body[5:5 + sn])
            ve = pb.VersionEntry.FromString(body[5 + sn:])
            if op == _CREATE:
                live[se.name] = (ve.dims, False)
            elif op == _DELETE:
                live.pop(se.name, None)
            elif op == _FREEZE and se.name in live:
                live[se.name] = (live[se.name][0], True)
        if good != len(data):  # torn tail
            with open(self._cat_path, "r+b") as f:
                f.truncate(good)
        return live

    def _catalog(self, op, name, dims):
        path = self._space_dir(name)
        se = pb.SpaceEntry(path=path, name=name).SerializeToString()
        ve = pb.VersionEntry(path=os.path.join(path, DEFAULT_VERSION), space=name, name=DEFAULT_VERSION,
                             dims=dims).SerializeToString()
        body = bytes([op]) + struct.pack("<I", len(se)) + se + ve
        with self._mu:
            self._cat.write(struct.pack("<I", len(body)) + body)
            self._cat.flush()
            os.fsync(self._cat.fileno())

    def _open(self, name, dims, fresh=False):
        if fresh:
            # a directory left behind by a crash between the catalog's DELETE record and the removal of the files
            # must not be adopted: its rows belong to the deleted space
            shutil.rmtree(self._space_dir(name), ignore_errors=True)
        inner = self._inner.create_space(name, dims)
        sp = DurableSpace(inner, _SpaceLog(self._space_dir(name), dims, self._sync), self, name)
        self._spaces[name] = sp
        return sp

    # ---- the store interface of server.py ----
    def create_space(self, name, dims):
        with self._mu:
            sp = self._spaces.get(name)
            if sp is not None:
                return sp
            self._catalog(_CREATE, name, dims)
            return self._open(name, dims, fresh=True)

    def get_space(self, name):
        with self._mu:
            return self._spaces.get(name)

    def delete_space(self, name):
        with self._mu:
            sp = self._spaces.pop(name, None)
            if sp is None:
                return
            self._catalog(_DELETE, name, sp.dims)
            with sp._mu:  # no write of this space is between its log append and its apply
                sp._log.close()
            shutil.rmtree(self._space_dir(name), ignore_errors=True)
            self._inner.delete_space(name)  # (handlers still holding the engine space get EHX_ENOTFOUND: tombstone)

    def close(self):
        with self._mu:
            for sp in self._spaces.values():
                sp._log.close()
            self._cat.close()
