"""HTTP front of the reference (vfs/src/main.rs, vfs/src/vfs/tcp.rs) over the vfsx engine:
the `/search` JSON contract and the routes around it, bug-compatible
where that is observable.

Kept from the reference
  * routes and status codes (tcp.rs:107-150), the framing
    `HTTP/1.1 <status>\\r\\nContent-Type: application/json\\r\\nContent-Length: N\\r\\n\\r\\n<body>`
    (tcp.rs:153-158) and the status table, including 422 being sent as "200 OK" (tcp.rs:169-179);
  * `/search` passes `top_k` as the scan batch size and `ef_search` (default 6) as the RESULT
    LIMIT (tcp.rs:397-403; README says the opposite -- SURVEY.md Q5);
  * bodies built with `json!` have alphabetically ordered keys and every f32 widened to f64
    (0.1f32 -> 0.10000000149011612); `GET /vectors/<id>` is `to_string` of a struct: declaration
    order, shortest f32 digits (tcp.rs:249-255); NaN/inf -> null;
  * one request at a time (global mutex, main.rs:49).
Deliberate deviations (documented in DESIGN.md)
  * the whole request is read (Content-Length), not one 4096-byte read;
  * `search_type` other than "exact" (the default is approximate/HNSW, tcp.rs:383-386) is served
    by the exact engine -- the HNSW path is out of scope;
  * a registered vector is searchable at once and appears once (the reference returns memtable
    vectors twice); a quantizing store shows the flushed form of a vector
    (name + "_quantized", tag "quantized", vector.rs:116-121) from the start.
"""
from __future__ import annotations

import ctypes as C
import json
import math
import os
import socketserver
import threading
import time
from typing import Optional, Tuple

import numpy as np

from . import _lib

STATUS_TEXT = {200: "200 OK", 201: "201 Created", 204: "204 No Content", 400: "400 Bad Request",
               404: "404 Not Found", 500: "500 Internal Server Error"}  # tcp.rs:169-179


# ---- number formatting (serde_json / ryu) ----------------------------------------------------

def _ryu(digits: str, exp10: int, lo: int, hi: int) -> str:
    """digits d1d2..dn with value 0.d1d2..dn * 10^exp10 (exp10 = ryu's kk), rendered by ryu's
    rules: plain decimals while lo < kk <= hi, exponent form otherwise."""
    n = len(digits)
    kk = exp10
    if n <= kk <= hi:
        return digits + "0" * (kk - n) + ".0"
    if 0 < kk <= hi:
        return digits[:kk] + "." + digits[kk:]
    if lo < kk <= 0:
        return "0." + "0" * (-kk) + digits
    e = kk - 1
    mant = digits[0] + ("." + digits[1:] if n > 1 else "")
    return f"{mant}e{e}"


def _split_repr(r: str) -> Tuple[str, str, int]:
    """'-1.25e-07' -> ('-', '125', kk) with value = 0.125 * 10^kk."""
    sign = ""
    if r[0] == "-":
        sign, r = "-", r[1:]
    mant, _, e = r.partition("e")
    exp = int(e) if e else 0
    ip, _, fp = mant.partition(".")
    digits = (ip + fp).lstrip("0")
    if not digits:
        return sign, "0", 1
    lead_zeros = len(ip + fp) - len((ip + fp).lstrip("0"))
    kk = len(ip) - lead_zeros + exp
    digits = digits.rstrip("0") or "0"
    return sign, digits, kk


def fmt_f64(x: float) -> str:
    """serde_json's rendering of an f64 (ryu, shortest round-trip); non-finite -> null."""
    if not math.isfinite(x):
        return "null"
    if x == 0:
        return "-0.0" if math.copysign(1.0, x) < 0 else "0.0"
    sign, digits, kk = _split_repr(repr(float(x)))
    return sign + _ryu(digits, kk, -5, 16)


def fmt_f32_widened(x) -> str:
    """An f32 that went through serde_json::Value (json! / to_value): widened to f64 first."""
    return fmt_f64(float(np.float32(x)))


def fmt_f32(x) -> str:
    """An f32 serialised directly (serde_json::to_string of a struct): shortest f32 digits."""
    v = np.float32(x)
    if not np.isfinite(v):
        return "null"
    if v == 0:
        return "-0.0" if np.signbit(v) else "0.0"
    sign, digits, kk = _split_repr(np.format_float_scientific(v, unique=True, trim="-"))
    return sign + _ryu(digits, kk, -6, 13)


def jstr(s: str) -> str:
    return json.dumps(s, ensure_ascii=False)


def jobj(pairs) -> str:
    """`json!` object: keys sorted (serde_json without preserve_order is a BTreeMap)."""
    return "{" + ",".join(f"{jstr(k)}:{v}" for k, v in sorted(pairs, key=lambda kv: kv[0])) + "}"


def err_body(msg: str) -> str:
    return jobj([("error", jstr(msg))])


# ---- the state behind the mutex (ServerState, tcp.rs:60-63) --------------------------------------


def _new_corpus(dim: int, quantize: bool, device: int):
    """The resident store.  VFSX_SERVER_MIRROR=1 switches the matrix-core path on (no reference counterpart; results
    unchanged): /search on an f32 store of >= 1.75e8 elements with ef_search <= 32 is then pre-filtered from the 1 B per
    element int8 mirror instead of scanning 4 B per element; inserts extend the mirror (vfsx_api.cpp ensure_mirror)."""
    import vfs_amd
    corpus = vfs_amd.Corpus(dim, quantize=quantize, device=device)
    if os.environ.get("VFSX_SERVER_MIRROR", "0") not in ("", "0"):
        corpus.enable_mfma(True)
    return corpus

FLUSH_THRESHOLD = 10  # storage_manager.rs:12
INSERT_BATCH = 4096   # one-row appends staged on the host before they are committed to the device together


class ServerState:
    def __init__(self, dim: int, quantize: bool, name: str, device: int = 0, state_dir: str = "state",
                 data_dir: str = "data"):
        self.vector_dimension = dim
        self.quantize = quantize
        self.name = name
        self.corpus = _new_corpus(dim, quantize, device)
        self.corpus.set_insert_batch(INSERT_BATCH)
        self._rows = 0                        # rows registered (local row of the next one)
        self.next_id = 1                      # storage_manager.rs:76
        self.row_of = {}                      # id -> local row
        self.meta = {}                        # id -> (name, tags) as the store holds them
        self.memtable = []                    # [(id, created_at)] registered, not yet in the log
        self.index_map = {}                   # id -> log offset (storage_manager.rs:61)
        self.current_offset = 0
        self.state_path = os.path.join(state_dir, "vfs_state.bin")   # storage_manager.rs:14
        self.data_path = os.path.join(data_dir, "vectors.dat")       # storage_manager.rs:13

    def register(self, values, name: str, tags) -> int:
        """register_vector_from_vec (storage_manager.rs:317-335): id = next_id++, into the memtable, flushed to the
        log at 10.  The vector goes to the engine at once through a one-row append that the library batches on the
        host (vfsx_corpus_set_insert_batch): it is searchable when this returns, the device work (H2D, rb64 layout /
        quantize, norm) happens once per batch -- at the next search, or at the flush below for a quantizing store,
        whose log records hold the stored int8 form."""
        from . import vfslog
        vid = self.next_id
        self.next_id += 1
        v = np.ascontiguousarray(values, dtype=np.float32)
        row = self._rows
        self._append(self.corpus.handle, v.ctypes.data, C.byref(C.c_uint64(vid)), 1)
        self._rows = row + 1
        self.row_of[vid] = row
        if self.quantize:  # Vector::quantize renames and tags the stored form (vector.rs:116-121)
            self.meta[vid] = (name + "_quantized", list(tags) + ["quantized"])
            self.memtable.append((vid, None, v))
        else:
            self.meta[vid] = (name, list(tags))
            self.memtable.append((vid, vfslog.utc_now_rfc3339(), v))
        if len(self.memtable) >= FLUSH_THRESHOLD:  # storage_manager.rs:310-312
            self.flush()
        return vid

    def _append(self, handle, data_ptr, id_ref, n) -> None:
        lib = self.corpus._lib
        fn = lib.vfsx_corpus_append_quantize if self.quantize else lib.vfsx_corpus_append
        rc = fn(handle, C.cast(data_ptr, C.POINTER(C.c_float)) if self.quantize else C.c_void_p(data_ptr), id_ref, n)
        if rc != _lib.OK:
            _lib.check(rc)

    def flush(self) -> None:
        """flush_memtable_to_disk (storage_manager.rs:139-157): append the pending vectors to the log in
        the reference's record format -- the stored form, i.e. the int8 values read back from HBM for a
        quantizing store -- and index them by offset."""
        from . import vfslog
        if not self.memtable:
            return
        recs = []
        if self.quantize:  # quantize() stamps a fresh created_at (vector.rs:122)
            # the stored int8 form, by the store's own rule on the host (vfsx_quantize_rows_host = the bytes the device
            # quantizer left in HBM: no read-back, no commit of the rows still staged in the insert batch)
            f32 = np.ascontiguousarray(np.stack([v for _, _, v in self.memtable]), dtype=np.float32)
            values = np.empty(f32.shape, dtype=np.int8)
            _lib.check(self.corpus._lib.vfsx_quantize_rows_host(
                f32.ctypes.data_as(C.POINTER(C.c_float)), f32.size, 127.0, values.ctypes.data_as(C.POINTER(C.c_int8))))
            now = vfslog.utc_now_rfc3339()
            for (vid, _, _), v in zip(self.memtable, values):
                name, tags = self.meta[vid]
                recs.append(vfslog.encode_record(vid, v, name, tags, now, scale=127.0))
        else:
            for vid, created, v in self.memtable:
                name, tags = self.meta[vid]
                recs.append(vfslog.encode_record(vid, v, name, tags, created))
        offsets = vfslog.append_records(self.data_path, recs)
        for (vid, _, _), off in zip(self.memtable, offsets):
            self.index_map[vid] = off
        self.memtable = []

    def load_log(self, path: str) -> int:
        row0 = len(self.corpus)
        n = self.corpus.load_log(path)
        self._rows = row0 + n
        ids = self.corpus.get_ids(np.arange(row0, row0 + n))
        for i, vid in enumerate(ids):
            self.row_of[int(vid)] = row0 + i
        if n:
            self.next_id = max(self.next_id, int(ids.max()) + 1)
        return n

    def vector(self, vid: int):
        row = self.row_of.get(vid)
        if row is None:
            return None
        if vid in self.meta:
            name, tags = self.meta[vid]
        else:
            m = self.corpus.get_metadata(row)
            name, tags = m["name"], m["tags"]
        return vid, self.corpus.get_rows_f32([row])[0], name, tags


class App:
    """Routes of tcp.rs:107-150 on one ServerState behind one lock."""

    def __init__(self, device: int = 0, state_dir: str = "state", data_dir: Optional[str] = None):
        self.lock = threading.Lock()
        self.state: Optional[ServerState] = None
        self.device = device
        self.state_dir = state_dir
        # the reference keeps data/ and state/ side by side under its working directory
        self.data_dir = data_dir if data_dir is not None else os.path.join(
            os.path.dirname(os.path.abspath(state_dir)), "data")

    # -- helpers -------------------------------------------------------------------------------
    @staticmethod
    def _vector_value(vid, values, name, tags) -> str:  # inside json!: sorted keys, widened floats
        return jobj([("id", str(vid)), ("name", jstr(name)), ("tags", "[" + ",".join(jstr(t) for t in tags) + "]"),
                     ("values", "[" + ",".join(fmt_f32_widened(v) for v in values) + "]")])

    # -- routes ----------------------------------------------------------------------------------
    def handle(self, method: str, path: str, body: str) -> Tuple[int, str]:
        with self.lock:
            try:
                return self._route(method, path, body)
            except Exception as e:  # a panic in the reference poisons the mutex; here: 500
                return 500, err_body(f"internal error: {e}")

    def _route(self, method: str, path: str, body: str) -> Tuple[int, str]:
        if method == "GET" and path.startswith("/vectors/"):
            ident = path[len("/vectors/"):]
            if not ident.isdigit():  # u64 parse, tcp.rs:110-113
                return 400, err_body("Invalid vector ID")
            return self.get_vector(int(ident))
        if method == "POST" and path == "/init":
            req = self._parse(body, {"vector_dimension": int, "truncate_data": bool, "quantize": bool})
            if req is None:
                return 400, err_body("Invalid JSON for init")
            return self.init(req)
        if method == "POST" and path == "/vectors":
            req = self._parse(body, {"values": list, "name": str, "tags": list})
            if req is None:
                return 400, err_body("Invalid JSON request")
            return self.register_vector(req)
        if method == "POST" and path == "/search":
            req = self._parse(body, {"values": list, "top_k": int})
            if req is None:
                return 400, err_body("Invalid JSON request")
            return self.search(req)
        if method == "POST" and path == "/flush":
            if self.state is None:
                return 400, err_body("VFSManager is not initialized")
            try:
                self.state.flush()
            except OSError as e:  # VFSError::MemtableError (tcp.rs:221-228)
                return 500, jobj([("error", jstr(f"Memtable error: I/O error: {e}")),
                                  ("error_type", jstr("memtable_error")), ("status", jstr("error"))])
            return 200, jobj([("status", jstr("success")), ("message", jstr("Memtable flushed to disk"))])
        if method == "POST" and path == "/snapshot":
            return self.snapshot()
        if method == "POST" and path == "/restore":
            return self.restore()
        return 404, err_body("Not found")

    @staticmethod
    def _parse(body: str, required: dict):
        try:
            req = json.loads(body)
        except (ValueError, TypeError):
            return None
        if not isinstance(req, dict):
            return None
        for key, typ in required.items():
            v = req.get(key)
            if typ is int:
                if isinstance(v, bool) or not isinstance(v, int) or v < 0:
                    return None
            elif typ is bool:
                if not isinstance(v, bool):
                    return None
            elif not isinstance(v, typ):
                return None
        if "values" in required and not all(isinstance(x, (int, float)) and not isinstance(x, bool)
                                            for x in req["values"]):
            return None
        if "tags" in required and not all(isinstance(x, str) for x in req["tags"]):
            return None
        return req

    def init(self, req) -> Tuple[int, str]:  # tcp.rs:270-299
        if self.state is not None:
            return 400, err_body("VFSManager is already initialized")
        name = req.get("storage_name") or "default_vfs"
        st = ServerState(req["vector_dimension"], req["quantize"], name, self.device, self.state_dir, self.data_dir)
        if req["truncate_data"]:  # reset_state(truncate_data_file), storage_manager.rs:85-98
            os.makedirs(os.path.dirname(st.data_path) or ".", exist_ok=True)
            open(st.data_path, "wb").close()
        else:
            # the reference's scan reads the log from offset 0 (rank.rs:139-156), so an existing data file
            # is part of every search: stream it into HBM now.  A file this store cannot hold (other
            # dimension or storage type) is left alone, as the reference would only fail at search time.
            self._load_existing_log(st)
        self.state = st
        return 200, jobj([("status", jstr("initialized"))])

    @staticmethod
    def _load_existing_log(st: ServerState) -> int:
        from . import vfslog, _lib
        if len(st.corpus) or not os.path.exists(st.data_path) or os.path.getsize(st.data_path) == 0:
            return 0
        try:
            info = vfslog.inspect(st.data_path)
            if info["records"] == 0 or info["dim"] != st.vector_dimension or info["quantized"] != st.quantize:
                return 0
            return st.load_log(st.data_path)
        except _lib.VfsxError:
            return 0

    def register_vector(self, req) -> Tuple[int, str]:  # tcp.rs:301-363
        st = self.state
        if st is None:
            return 400, err_body("VFSManager is not initialized")
        if len(req["values"]) != st.vector_dimension:
            return 400, err_body(f"Vector dimension mismatch. Expected {st.vector_dimension}, "
                                 f"got {len(req['values'])}")
        if not req["values"]:  # storage_manager.rs:319-321 -> VFSError::InvalidVector -> 400
            return 400, jobj([("error", jstr("Invalid vector: Vector data cannot be empty")),
                              ("error_type", jstr("invalid_vector"))])
        vid = st.register(req["values"], req["name"], req["tags"])
        return 201, jobj([("id", str(vid)), ("status", jstr("success"))])

    def get_vector(self, vid: int) -> Tuple[int, str]:  # tcp.rs:242-268
        st = self.state
        if st is None:
            return 400, err_body("VFSManager is not initialized")
        v = st.vector(vid)
        if v is None:
            return 404, err_body("Vector not found")
        _, values, name, tags = v
        # serde_json::to_string(&VectorResponse): declaration order id, values, name, tags; f32 digits
        return 200, ("{" + f"\"id\":{vid},\"values\":[" + ",".join(fmt_f32(x) for x in values) + "],"
                     + f"\"name\":{jstr(name)},\"tags\":[" + ",".join(jstr(t) for t in tags) + "]}")

    def search(self, req) -> Tuple[int, str]:  # tcp.rs:366-450
        import vfs_amd
        st = self.state
        if st is None:
            return 400, err_body("VFSManager is not initialized")
        if len(req["values"]) != st.vector_dimension:
            return 400, err_body(f"Query vector dimension mismatch. Expected {st.vector_dimension}, "
                                 f"got {len(req['values'])}")
        method = vfs_amd.DistanceMethod.from_request(req.get("distance_method")
                                                     if isinstance(req.get("distance_method"), str) else None)
        ef = req.get("ef_search")
        ef_search = ef if isinstance(ef, int) and not isinstance(ef, bool) and ef >= 0 else 6
        ranker = vfs_amd.Ranker(vfs_amd.SearchType.Exact, method, ef_search)
        t0 = time.perf_counter()
        try:  # ranker.search(&query, req.top_k /*batch*/, Some(ef_search) /*limit*/, manager): tcp.rs:403
            hits = ranker.search(req["values"], req["top_k"], ef_search, st.corpus)
        except vfs_amd.VfsxError as e:
            return 500, err_body(f"Search error: {e}")
        ms = np.float32(np.float32(time.perf_counter() - t0) * np.float32(1000.0))
        results = []
        for vid, dist in hits:
            v = st.vector(vid)
            vec = "null" if v is None else self._vector_value(*v)
            results.append(jobj([("id", str(vid)), ("distance", fmt_f32_widened(dist)), ("vector", vec)]))
        return 200, jobj([("results", "[" + ",".join(results) + "]"), ("query_time_ms", fmt_f32_widened(ms))])

    def snapshot(self) -> Tuple[int, str]:  # tcp.rs:182-195, storage_manager.rs:367-410
        """Writes state/vfs_state.bin in the reference's bincode layout and flushes the memtable.
        Deviation: the reference encodes the state BEFORE the flush, so its snapshot's index_map misses
        the vectors that flush writes; here the flush comes first and the snapshot indexes everything."""
        from . import vfslog
        st = self.state
        if st is None:
            return 400, err_body("VFS Manager not initialized")
        try:
            st.flush()
            os.makedirs(os.path.dirname(st.state_path) or ".", exist_ok=True)
            with open(st.state_path, "wb") as f:
                f.write(vfslog.encode_state(st.name, st.next_id, st.current_offset, st.index_map, st.quantize))
        except OSError as e:
            return 500, err_body(f"Failed to save state: I/O error: {e}")
        return 200, jobj([("status", jstr("State saved successfully"))])

    def restore(self) -> Tuple[int, str]:  # tcp.rs:198-211, storage_manager.rs:412-427
        """Loads the VFSState; the vectors themselves are re-streamed from the log into HBM when this
        store is still empty (the reference keeps them on disk and re-reads the log on every search)."""
        from . import vfslog
        st = self.state
        if st is None:
            return 400, err_body("VFS Manager not initialized")
        try:
            with open(st.state_path, "rb") as f:
                saved = vfslog.decode_state(f.read())
        except OSError as e:
            return 500, err_body(f"Failed to load state: I/O error: {e}")
        except ValueError as e:
            return 500, err_body(f"Failed to load state: Serialization error: {e}")
        if saved["quantize"] != st.quantize and len(st.corpus):
            return 500, err_body("Failed to load state: Internal error: snapshot storage type differs from the "
                                 "resident corpus")
        if saved["quantize"] != st.quantize:  # empty store: adopt the snapshot's storage type
            st.corpus.close()
            st.quantize = saved["quantize"]
            st.corpus = _new_corpus(st.vector_dimension, st.quantize, self.device)
        st.name = saved["name"]
        st.index_map = dict(saved["index_map"])
        st.current_offset = saved["current_offset"]
        self._load_existing_log(st)
        st.next_id = max(st.next_id, int(saved["next_id"]))
        return 200, jobj([("status", jstr("State loaded successfully"))])


# ---- socket layer (main.rs:47-62, tcp.rs:66-104,153-162) -----------------------------------------

def frame(status: int, body: str) -> bytes:
    b = body.encode("utf-8")
    head = (f"HTTP/1.1 {STATUS_TEXT.get(status, '200 OK')}\r\nContent-Type: application/json\r\n"
            f"Content-Length: {len(b)}\r\n\r\n")
    return head.encode("ascii") + b


def parse_request(raw: bytes) -> Optional[Tuple[str, str, str]]:
    text = raw.decode("utf-8", "replace")
    lines = text.splitlines()
    if not lines:
        return None
    parts = lines[0].split()
    if len(parts) < 2:
        return None
    # the body is every line after the first empty one, newlines dropped (tcp.rs:95-104)
    body, seen_blank = [], False
    for ln in lines:
        if seen_blank:
            body.append(ln)
        elif ln == "":
            seen_blank = True
    return parts[0], parts[1], "".join(body)


class _Handler(socketserver.BaseRequestHandler):
    def handle(self):
        sock = self.request
        sock.settimeout(10.0)
        data = b""
        try:
            while b"\r\n\r\n" not in data and b"\n\n" not in data:
                chunk = sock.recv(65536)
                if not chunk:
                    break
                data += chunk
            head, sep, rest = data.partition(b"\r\n\r\n")
            want = 0
            for ln in head.split(b"\r\n")[1:]:
                if ln.lower().startswith(b"content-length:"):
                    try:
                        want = int(ln.split(b":", 1)[1].strip())
                    except ValueError:
                        want = 0
            while sep and len(rest) < want:
                chunk = sock.recv(65536)
                if not chunk:
                    break
                rest += chunk
                data += chunk
        except OSError:
            return
        req = parse_request(data)
        if req is None:
            return
        status, body = self.server.app.handle(*req)
        try:
            sock.sendall(frame(status, body))
        except OSError:
            pass


class VfsServer(socketserver.ThreadingTCPServer):
    allow_reuse_address = True
    daemon_threads = True

    def __init__(self, port: int = 9001, host: str = "127.0.0.1", device: int = 0, state_dir: str = "state"):
        super().__init__((host, port), _Handler)
        self.app = App(device=device, state_dir=state_dir)


def main(argv=None) -> int:
    import sys
    argv = sys.argv[1:] if argv is None else argv
    # the documented invocation is `cargo run +nightly <port>` (main.rs:45-46): accept both shapes
    args = [a for a in argv if not a.startswith("+")]
    port = int(args[0]) if args else 9001
    srv = VfsServer(port=port)
    print(f"Servidor escuchando en http://127.0.0.1:{port}", flush=True)
    try:
        srv.serve_forever()
    except KeyboardInterrupt:
        pass
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
