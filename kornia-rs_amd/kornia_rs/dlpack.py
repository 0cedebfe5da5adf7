"""Zero-copy DLPack interop (producer + consumer) in pure ctypes.

Mirrors ``crates/kornia-tensor/src/dlpack.rs`` / ``crates/kornia-image/src/dlpack.rs`` and the
capsule glue of ``kornia-py/src/dlpack.rs``: export keeps the tensor alive through the managed
tensor's ``manager_ctx`` and frees it in the deleter (dlpack.rs:72-170); import accepts only
C-contiguous tensors of a known dtype with ``lanes == 1`` (:172-290).  Device codes are the ROCm
ones — Host -> kDLCPU, Device -> **kDLROCM (10)**, pinned -> kDLROCMHost (11), managed ->
kDLCUDAManaged (13) — because that is what torch-ROCm speaks; the reference's kDLCUDA would be
rejected by it.
"""
from __future__ import annotations

import ctypes as C
from typing import Any, Optional, Tuple

import numpy as np

kDLCPU, kDLCUDA, kDLCUDAHost, kDLROCM, kDLROCMHost, kDLCUDAManaged = 1, 2, 3, 10, 11, 13
_DEVICE_TYPES = (kDLCUDA, kDLROCM)           # device-resident for our purposes
_HOST_TYPES = (kDLCPU, kDLCUDAHost, kDLROCMHost)

kDLInt, kDLUInt, kDLFloat = 0, 1, 2
_CODE = {"u": kDLUInt, "i": kDLInt, "f": kDLFloat}
_KIND = {v: k for k, v in _CODE.items()}
SUPPORTED = {"uint8", "uint16", "int32", "int64", "float16", "float32", "float64"}  # DlpackElem, dlpack.rs:25


class DLDevice(C.Structure):
    _fields_ = [("device_type", C.c_int32), ("device_id", C.c_int32)]


class DLDataType(C.Structure):
    _fields_ = [("code", C.c_uint8), ("bits", C.c_uint8), ("lanes", C.c_uint16)]


class DLTensor(C.Structure):
    _fields_ = [("data", C.c_void_p), ("device", DLDevice), ("ndim", C.c_int32), ("dtype", DLDataType),
                ("shape", C.POINTER(C.c_int64)), ("strides", C.POINTER(C.c_int64)), ("byte_offset", C.c_uint64)]


class DLManagedTensor(C.Structure):
    pass


_DELETER = C.CFUNCTYPE(None, C.POINTER(DLManagedTensor))
DLManagedTensor._fields_ = [("dl_tensor", DLTensor), ("manager_ctx", C.c_void_p), ("deleter", _DELETER)]

_py = C.pythonapi
_py.PyCapsule_New.restype = C.py_object
_py.PyCapsule_New.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p]
_py.PyCapsule_GetPointer.restype = C.c_void_p
_py.PyCapsule_GetPointer.argtypes = [C.py_object, C.c_char_p]
_py.PyCapsule_IsValid.restype = C.c_int
_py.PyCapsule_IsValid.argtypes = [C.py_object, C.c_char_p]
_py.PyCapsule_SetName.restype = C.c_int
_py.PyCapsule_SetName.argtypes = [C.py_object, C.c_char_p]

# The capsule destructor runs while the capsule is being deallocated (refcount 0): it must not
# create a py_object reference to it, so it uses raw-pointer prototypes of the same C API.
_raw_is_valid = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_char_p)(("PyCapsule_IsValid", _py))
_raw_get_pointer = C.CFUNCTYPE(C.c_void_p, C.c_void_p, C.c_char_p)(("PyCapsule_GetPointer", _py))

_NAME, _USED = b"dltensor", b"used_dltensor"
_live: dict = {}  # address of an exported DLManagedTensor -> (struct, shape array, owner)
export_count = 0
release_count = 0


@_DELETER
def _deleter(mt_ptr):
    """Runs when the consumer is done: drop our keepalive (the exported tensor)."""
    global release_count
    addr = C.addressof(mt_ptr.contents)
    if _live.pop(addr, None) is not None:
        release_count += 1


def _neutralise_live_exports():
    """atexit: a consumer may release an exported tensor AFTER the interpreter has begun to finalise (a torch tensor that is
    still alive at shutdown was seen to crash the process with SIGSEGV in the deleter callback, profiles/r02a_runtimes.log).
    Every tensor still exported gets the C no-op deleter, and its struct / shape / owner are deliberately leaked so the
    consumer's pointer stays valid until the process ends."""
    from ._ffi import lib
    noop = C.cast(lib.kh_dlpack_noop_deleter, _DELETER)
    for mt, shp, owner in list(_live.values()):
        mt.deleter = noop
        for obj in (mt, shp, owner):
            C.pythonapi.Py_IncRef(C.py_object(obj))


import atexit  # noqa: E402

atexit.register(_neutralise_live_exports)

_CAPSULE_DTOR = C.CFUNCTYPE(None, C.c_void_p)


@_CAPSULE_DTOR
def _capsule_destructor(capsule_ptr):
    # A capsule that was never consumed still owns the managed tensor.
    if _raw_is_valid(capsule_ptr, _NAME):
        ptr = _raw_get_pointer(capsule_ptr, _NAME)
        mt = C.cast(ptr, C.POINTER(DLManagedTensor))
        if mt.contents.deleter:
            mt.contents.deleter(mt)


def export(owner: Any, data_ptr: int, shape: Tuple[int, ...], dtype: np.dtype, device_type: int,
           device_id: int) -> Any:
    """Build a ``dltensor`` capsule aliasing ``data_ptr``; ``owner`` stays alive until the
    consumer calls the deleter."""
    global export_count
    dtype = np.dtype(dtype)
    if dtype.name not in SUPPORTED:
        raise BufferError(f"dtype {dtype.name} has no DLPack mapping")
    shp = (C.c_int64 * len(shape))(*shape)
    mt = DLManagedTensor()
    mt.dl_tensor.data = data_ptr
    mt.dl_tensor.device = DLDevice(device_type, device_id)
    mt.dl_tensor.ndim = len(shape)
    mt.dl_tensor.dtype = DLDataType(_CODE[dtype.kind], dtype.itemsize * 8, 1)
    mt.dl_tensor.shape = C.cast(shp, C.POINTER(C.c_int64))
    mt.dl_tensor.strides = None  # NULL = compact row-major
    mt.dl_tensor.byte_offset = 0
    mt.manager_ctx = None
    mt.deleter = _deleter
    _live[C.addressof(mt)] = (mt, shp, owner)
    export_count += 1
    return _py.PyCapsule_New(C.addressof(mt), _NAME, C.cast(_capsule_destructor, C.c_void_p))


class _Imported:
    """Keepalive for an imported tensor: calls the producer's deleter exactly once."""

    def __init__(self, mt_ptr):
        self._mt = mt_ptr

    def __del__(self):
        mt, self._mt = self._mt, None
        if mt is not None and mt.contents.deleter:
            mt.contents.deleter(mt)


def import_capsule(capsule: Any):
    """-> (data_ptr, shape, np.dtype, device_type, device_id, keepalive).  Raises on non-contiguous
    strides, vector lanes or an unsupported dtype (tensor_from_dlpack_raw, T/dlpack.rs:172-265)."""
    if not _py.PyCapsule_IsValid(capsule, _NAME):
        raise ValueError("expected an unconsumed 'dltensor' capsule")
    ptr = _py.PyCapsule_GetPointer(capsule, _NAME)
    mt = C.cast(ptr, C.POINTER(DLManagedTensor))
    _py.PyCapsule_SetName(capsule, _USED)  # we own it now
    keep = _Imported(mt)
    t = mt.contents.dl_tensor
    if t.dtype.lanes != 1:
        raise ValueError(f"DLPack vector types are not supported (lanes = {t.dtype.lanes})")
    kind = _KIND.get(t.dtype.code)
    if kind is None or t.dtype.bits % 8:
        raise ValueError(f"unsupported DLPack dtype code {t.dtype.code} bits {t.dtype.bits}")
    dtype = np.dtype(f"{kind}{t.dtype.bits // 8}")
    if dtype.name not in SUPPORTED:
        raise ValueError(f"unsupported DLPack dtype {dtype.name}")
    shape = tuple(int(t.shape[i]) for i in range(t.ndim))
    if t.strides:
        expect = 1
        for i in range(t.ndim - 1, -1, -1):
            if shape[i] != 1 and int(t.strides[i]) != expect:
                raise ValueError("only C-contiguous DLPack tensors can be imported")
            expect *= shape[i]
    return int(t.data or 0) + int(t.byte_offset), shape, dtype, int(t.device.device_type), int(t.device.device_id), keep


def from_object(obj: Any, stream_ptr: Optional[int] = None):
    """Call ``obj.__dlpack__`` the way the array-API protocol prescribes and import the capsule."""
    if hasattr(obj, "__dlpack_device__"):
        dev_type, _ = obj.__dlpack_device__()
        if dev_type in _DEVICE_TYPES or dev_type == kDLCUDAManaged:
            # Managed memory is written by kernels too: its producer needs the consumer stream for the hand-off fence just like a
            # device producer.  ROCm convention: stream 0 = the default stream; we pass our consumer stream handle.
            capsule = obj.__dlpack__(stream=stream_ptr if stream_ptr else None)
        else:
            capsule = obj.__dlpack__()
    elif hasattr(obj, "__dlpack__"):
        capsule = obj.__dlpack__()
    else:
        capsule = obj  # already a capsule
    return import_capsule(capsule)
