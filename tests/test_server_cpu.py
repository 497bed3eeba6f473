"""CPU: the HTTP contract pieces of vfs_amd/server.py that need no GPU -- number rendering
(serde_json/ryu), framing, body extraction and the pre-init / error routes (tcp.rs)."""
import numpy as np

from vfs_amd import server


def test_f64_rendering_matches_serde_json():
    f = server.fmt_f64
    assert f(float(np.float32(0.1))) == "0.10000000149011612"
    assert f(3.0) == "3.0" and f(123.456) == "123.456" and f(-2.5) == "-2.5"
    assert f(1e-7) == "1e-7" and f(1.5e-7) == "1.5e-7"
    assert f(0.00001) == "0.00001" and f(0.000001) == "1e-6"   # ryu: plain while -5 < kk
    assert f(1e15) == "1000000000000000.0" and f(1e16) == "1e16" and f(1.5e16) == "1.5e16"
    assert f(123456789.0) == "123456789.0"
    assert f(float("nan")) == "null" and f(float("inf")) == "null"
    assert f(0.0) == "0.0"
    assert server.fmt_f32_widened(np.sqrt(np.float32(14.0))) == "3.7416574954986572"


def test_f32_rendering_shortest_digits():
    f = server.fmt_f32
    assert f(0.1) == "0.1" and f(1.0) == "1.0" and f(-0.007874016) == "-0.007874016"
    assert f(1e-6) == "0.000001" and f(1e-7) == "1e-7"        # ryu f32: plain while -6 < kk
    assert f(1e13) == "1e13" and f(1e12) == "1000000000000.0"
    assert f(np.float32(1) / np.float32(127)) == "0.007874016"


def test_json_object_is_key_sorted_and_compact():
    assert server.jobj([("results", "[]"), ("query_time_ms", "1.5")]) == '{"query_time_ms":1.5,"results":[]}'
    assert server.err_body('a "b"') == '{"error":"a \\"b\\""}'


def test_framing_and_status_table():
    raw = server.frame(201, '{"id":1}')
    assert raw == b'HTTP/1.1 201 Created\r\nContent-Type: application/json\r\nContent-Length: 8\r\n\r\n{"id":1}'
    assert server.frame(422, "{}").startswith(b"HTTP/1.1 200 OK\r\n")  # tcp.rs:177: unknown -> "200 OK"


def test_request_parsing_drops_newlines_in_body():
    raw = b'POST /init HTTP/1.1\r\nHost: x\r\nContent-Length: 9\r\n\r\n{"a":\n 1}\n'
    assert server.parse_request(raw) == ("POST", "/init", '{"a": 1}')


def test_routes_before_init_and_errors():
    app = server.App()
    assert app.handle("GET", "/vectors/1", "") == (400, '{"error":"VFSManager is not initialized"}')
    assert app.handle("GET", "/vectors/abc", "") == (400, '{"error":"Invalid vector ID"}')
    assert app.handle("POST", "/search", "not json") == (400, '{"error":"Invalid JSON request"}')
    assert app.handle("POST", "/search", '{"values":[1,2],"top_k":3}') == (
        400, '{"error":"VFSManager is not initialized"}')
    assert app.handle("POST", "/vectors", '{"values":[1],"name":"x"}') == (400, '{"error":"Invalid JSON request"}')
    assert app.handle("POST", "/init", '{"vector_dimension":4}') == (400, '{"error":"Invalid JSON for init"}')
    assert app.handle("POST", "/flush", "") == (400, '{"error":"VFSManager is not initialized"}')
    assert app.handle("POST", "/snapshot", "") == (400, '{"error":"VFS Manager not initialized"}')
    assert app.handle("DELETE", "/vectors/1", "") == (404, '{"error":"Not found"}')
