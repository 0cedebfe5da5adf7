"""Lookup-table cache and scratch workspaces of the operators that need them:
eviction never invalidates results, a cache miss or a pool allocation under stream capture is a typed error, and with a
registered workspace the same operators capture and replay."""
import ctypes as C

import numpy as np
import pytest

import oracle_ffi as O
from gpu_util import dev, out_buf

pytestmark = pytest.mark.gpu


def _resize_u8(stream, src, dw, dh, mode):
    from kornia_rs import _ffi
    h, w, c = src.shape
    s, d = dev(stream, src), out_buf(stream, dw * dh * c)
    _ffi.check(_ffi.lib.kh_resize_fast_u8(stream.cuda_stream_ptr, s.ptr, d.ptr, w, h, dw, dh, c, mode, 1, 1, 0, 0))
    return d.to_numpy(np.uint8, (dh, dw, c))


def test_table_cache_eviction_keeps_results_right(gpu_stream):
    """More distinct geometries than the cache holds (256 tables, two per call): early geometries are evicted and rebuilt, every
    result still equals the restatement."""
    from kornia_rs import _ffi
    src = O.pattern_u8(40 * 30 * 3).reshape(30, 40, 3)
    sizes = [(7 + (i % 180), 5 + (i // 3) % 90) for i in range(330)]
    for i, (dw, dh) in enumerate(sizes):
        got = _resize_u8(gpu_stream, src, dw, dh, _ffi.KH_INTERP_LANCZOS)
        if i % 11 == 0 or i >= 300:
            assert np.array_equal(got, O.resize_fast_u8(src, dw, dh, "lanczos", True)[0]), (i, dw, dh)
    for dw, dh in sizes[:5]:  # evicted long ago: rebuilt
        assert np.array_equal(_resize_u8(gpu_stream, src, dw, dh, _ffi.KH_INTERP_LANCZOS), O.resize_fast_u8(src, dw, dh, "lanczos", True)[0])


def test_capture_needs_warm_tables_and_a_workspace():
    from kornia_rs import _ffi, hip
    from kornia_rs.hip import DeviceBuffer, Graph
    stream = hip.Stream.new(0)  # a stream of our own: the capture state must not leak into other tests
    src = O.pattern_u8(64 * 48 * 3).reshape(48, 64, 3)
    dw, dh = 23, 17  # a geometry no other test uses
    s, d = dev(stream, src), out_buf(stream, dw * dh * 3)

    def call():
        _ffi.check(_ffi.lib.kh_resize_fast_u8(stream.cuda_stream_ptr, s.ptr, d.ptr, 64, 48, dw, dh, 3, _ffi.KH_INTERP_BICUBIC, 1, 1, 0, 0))

    with pytest.raises(_ffi.KorniaHipError) as e:  # cold table: building it allocates + copies synchronously
        Graph.capture(call, [s, d], stream)
    assert e.value.code == _ffi.KH_ERR_INVALID_ARG and "before kh_graph_capture_begin" in str(e.value)
    call()  # warm-up outside capture: builds the tables, takes pool scratch
    need = hip.last_workspace_bytes()
    assert need == 2 * dw * 3 * 48
    with pytest.raises(_ffi.KorniaHipError) as e:  # warm tables, but the intermediate would come from the pool
        Graph.capture(call, [s, d], stream)
    assert "kh_stream_set_workspace" in str(e.value) and str(need) in str(e.value)
    ws = DeviceBuffer(need, stream, zeroed=False)
    stream.set_workspace(ws)
    try:
        _ffi.check(_ffi.lib.kh_memset_async(d.ptr, 0, dw * dh * 3, stream.cuda_stream_ptr))
        g = Graph.capture(call, [s, d, ws], stream)
        stream.synchronize()
        assert not d.to_numpy(np.uint8, (dh, dw, 3)).any()  # captured, not run
        g.replay()
        assert np.array_equal(d.to_numpy(np.uint8, (dh, dw, 3)), O.resize_fast_u8(src, dw, dh, "bicubic", True)[0])
    finally:
        stream.set_workspace(None)
    call()  # unregistered again: back to the pool
    assert np.array_equal(d.to_numpy(np.uint8, (dh, dw, 3)), O.resize_fast_u8(src, dw, dh, "bicubic", True)[0])


def test_u8_warps_with_a_workspace_capture_and_match(gpu_stream):
    from kornia_rs import _ffi, hip
    from kornia_rs.hip import DeviceBuffer, Graph
    stream = hip.Stream.new(0)
    src = O.pattern_u8(97 * 61 * 3).reshape(61, 97, 3)
    m = (C.c_float * 6)()
    _ffi.lib.kh_get_rotation_matrix2d(48.0, 30.0, 12.0, 0.9, m)
    s, d = dev(stream, src), out_buf(stream, 97 * 61 * 3)

    def call():
        _ffi.check(_ffi.lib.kh_warp_affine_u8(stream.cuda_stream_ptr, s.ptr, d.ptr, 97, 61, 97, 61, 3, m, 1, 0, 0))

    call()
    want = O.warp_affine_u8(src, np.array(list(m), np.float32), 97, 61)
    assert np.array_equal(d.to_numpy(np.uint8, (61, 97, 3)), want)
    need = hip.last_workspace_bytes()
    assert need > 0
    with pytest.raises(_ffi.KorniaHipError):
        Graph.capture(call, [s, d], stream)
    ws = DeviceBuffer(need, stream, zeroed=False)
    stream.set_workspace(ws)
    try:
        _ffi.check(_ffi.lib.kh_memset_async(d.ptr, 0, 97 * 61 * 3, stream.cuda_stream_ptr))
        g = Graph.capture(call, [s, d, ws], stream)
        for _ in range(3):
            g.replay()
        assert np.array_equal(d.to_numpy(np.uint8, (61, 97, 3)), want)
    finally:
        stream.set_workspace(None)


def test_workspace_argument_errors():
    from kornia_rs import _ffi
    assert _ffi.lib.kh_stream_set_workspace(None, None, 16) == _ffi.KH_ERR_INVALID_ARG
    assert _ffi.lib.kh_stream_set_workspace(None, 4096, 0) == _ffi.KH_ERR_INVALID_ARG
    assert _ffi.lib.kh_stream_set_workspace(None, None, 0) == _ffi.KH_OK
    assert _ffi.lib.kh_last_workspace_bytes(None) == _ffi.KH_ERR_INVALID_ARG


def _registered(stream) -> int:
    from kornia_rs import _ffi
    n = C.c_size_t(0)
    _ffi.check(_ffi.lib.kh_stream_workspace_bytes(stream.cuda_stream_ptr, C.byref(n)))
    return int(n.value)


def test_workspace_outlives_a_temporary_stream_wrapper():
    """`Stream.from_handle(h).set_workspace(buf)` on a temporary wrapper used to drop the only reference to `buf`
    while the C registry still pointed at it.  The registration itself now keeps the buffer alive."""
    import gc
    from kornia_rs import hip
    from kornia_rs.hip import DeviceBuffer
    stream = hip.Stream.new(0)
    hip.Stream.from_handle(stream.cuda_stream_ptr, 0).set_workspace(DeviceBuffer(4096, stream, zeroed=False))
    gc.collect()
    key = (0, stream.cuda_stream_ptr)
    assert key in hip._WORKSPACES and hip._WORKSPACES[key].ptr != 0  # still allocated
    assert _registered(stream) == 4096
    stream.set_workspace(None)
    assert key not in hip._WORKSPACES and _registered(stream) == 0


def test_freeing_a_registered_workspace_unregisters_it():
    from kornia_rs import hip
    from kornia_rs.hip import DeviceBuffer
    stream = hip.Stream.new(0)
    ws = DeviceBuffer(8192, stream, zeroed=False)
    stream.set_workspace(ws)
    assert _registered(stream) == 8192
    ws.free()  # the library must not keep a pointer to freed memory
    assert _registered(stream) == 0 and (0, stream.cuda_stream_ptr) not in hip._WORKSPACES
    # replacing a registration releases the old buffer's bookkeeping
    a, b = DeviceBuffer(1024, stream, zeroed=False), DeviceBuffer(2048, stream, zeroed=False)
    stream.set_workspace(a)
    stream.set_workspace(b)
    assert _registered(stream) == 2048 and not a._ws_keys and b._ws_keys
    stream.set_workspace(None)


def test_destroying_a_stream_drops_its_workspace():
    """A later stream may reuse the handle value: it must not inherit the stale entry."""
    from kornia_rs import _ffi, hip
    from kornia_rs.hip import DeviceBuffer
    h = C.c_void_p(0)
    _ffi.check(_ffi.lib.kh_stream_create(C.byref(h)))
    ws = DeviceBuffer(4096, hip.Stream.default(0), zeroed=False)
    _ffi.check(_ffi.lib.kh_stream_set_workspace(h, ws.ptr, 4096))
    n = C.c_size_t(0)
    _ffi.check(_ffi.lib.kh_stream_workspace_bytes(h, C.byref(n)))
    assert n.value == 4096
    _ffi.check(_ffi.lib.kh_stream_destroy(h))
    _ffi.check(_ffi.lib.kh_stream_workspace_bytes(h, C.byref(n)))  # a map lookup by value; the handle is not dereferenced
    assert n.value == 0


def test_a_dropped_stream_and_its_workspace_are_collected():
    """the usual pattern `ws = DeviceBuffer(n, s); s.set_workspace(ws)` made a module-global table pin ws -> s for the
    life of the process, so a service that creates a stream + workspace per request leaked both.  An owned stream now holds its
    workspace itself: dropping the two objects leaves a plain cycle the collector reclaims (stream destroyed, buffer freed)."""
    import gc
    import weakref
    from kornia_rs import hip
    from kornia_rs.hip import DeviceBuffer
    refs = []
    for _ in range(8):
        s = hip.Stream.new(0)
        ws = DeviceBuffer(1 << 20, s, zeroed=False)
        s.set_workspace(ws)
        assert _registered(s) == 1 << 20 and (0, s.cuda_stream_ptr) not in hip._WORKSPACES
        refs.append((weakref.ref(s), weakref.ref(ws)))
        del s, ws
    gc.collect()
    assert all(rs() is None and rw() is None for rs, rw in refs)
