"""ctypes binding of the C ABI in include/nnn_batch.h and include/rnnoise.h."""
import ctypes as C
import os

import numpy as np

FRAME_SIZE = 480

TAPS = ["filtered", "xlp", "ac", "lpc2", "xcorr1", "best1", "xcorr2c", "pitch_search", "pitch", "pitch_gain",
        "X", "P", "ex", "ep", "exp", "features", "silence", "g_raw", "g", "vad", "branch"]

# every symbol the two headers declare (tests check that the built library exports all of them)
BATCH_SYMBOLS = [
    "nnn_model_from_bytes", "nnn_model_default", "nnn_model_free", "nnn_model_clone", "nnn_model_shape",
    "nnn_convert_rnnoise_text", "nnn_model_from_rnnoise_text",
    "nnn_batch_create", "nnn_batch_create_grouped", "nnn_batch_destroy", "nnn_batch_num_streams", "nnn_batch_reset",
    "nnn_batch_process_device", "nnn_batch_process_host", "nnn_batch_process_pcm_device", "nnn_batch_process_pcm_host",
    "nnn_batch_synchronize", "nnn_batch_clone", "nnn_batch_state_bytes", "nnn_batch_save_state", "nnn_batch_load_state",
    "nnn_batch_set_taps", "nnn_batch_set_schedule", "nnn_batch_set_inputs_ready", "nnn_debug_activations",
    "nnn_tap_info", "nnn_batch_read_tap", "nnn_batch_set_profiling", "nnn_batch_num_kernels",
    "nnn_batch_kernel_name", "nnn_batch_read_kernel_times", "nnn_batch_set_graph", "nnn_batch_set_pipeline", "nnn_batch_read_stamps",
    "nnn_host_alloc", "nnn_host_free", "nnn_last_error", "nnn_batch_fault", "nnn_batch_debug_withhold_flag", "nnn_batch_set_frame_log",
    "nnn_batch_create_opts", "nnn_batch_max_group_frames", "nnn_batch_device_bytes", "nnn_batch_set_back_end", "nnn_device_local_cpulist",
]
TRAIN_SYMBOLS = [
    "nnn_train_create", "nnn_train_destroy", "nnn_train_reset", "nnn_train_process_device", "nnn_train_process_host",
]
RESAMPLE_SYMBOLS = [
    "nnn_resampler_create", "nnn_resampler_destroy", "nnn_resampler_reset", "nnn_resampler_max_output",
    "nnn_resampler_process_device", "nnn_resampler_process_host",
]
NODE_SYMBOLS = [
    "nnn_node_create", "nnn_node_destroy", "nnn_node_num_streams", "nnn_node_num_shards", "nnn_node_shard", "nnn_node_batch", "nnn_node_reset",
    "nnn_node_process_host", "nnn_node_process_pcm_host", "nnn_node_process_device", "nnn_node_process_device_streams", "nnn_node_synchronize",
    "nnn_node_fault", "nnn_node_shard_cpus",
]
RNNOISE_SYMBOLS = [
    "rnnoise_get_frame_size", "rnnoise_get_size", "rnnoise_init", "rnnoise_create", "rnnoise_destroy",
    "rnnoise_process_frame", "rnnoise_model_from_file", "rnnoise_model_free",
]


PCM_F32, PCM_I16, PCM_F32_UNIT = 0, 1, 2
PCM_DTYPE = {PCM_F32: np.float32, PCM_I16: np.int16, PCM_F32_UNIT: np.float32}


class PcmLayout(C.Structure):
    """struct nnn_pcm_layout (include/nnn_batch.h)."""
    _fields_ = [("format", C.c_int), ("channels", C.c_int), ("discard_first", C.c_int), ("reserved", C.c_int),
                ("group_stride", C.c_size_t), ("frame_stride", C.c_size_t)]


class BatchOpts(C.Structure):
    """struct nnn_batch_opts (include/nnn_batch.h)."""
    _fields_ = [("max_group_frames", C.c_int), ("reserved", C.c_int * 7)]


class Library:
    """A loaded build of the backend.  The package default is the hipcc-built gfx950 library."""

    def __init__(self, path):
        if not os.path.exists(path):
            raise RuntimeError(
                f"nnnoiseless_amd: HIP library not found at {path}. Build it with "
                "`python -c 'import __graft_entry__ as g; g.build()'` (needs hipcc); there is no CPU fallback.")
        self.path = path
        L = self.L = C.CDLL(path)
        vp, sz, i32 = C.c_void_p, C.c_size_t, C.c_int
        L.nnn_model_from_bytes.restype = vp
        L.nnn_model_from_bytes.argtypes = [C.c_char_p, sz]
        L.nnn_model_default.restype = vp
        L.nnn_model_free.argtypes = [vp]
        L.nnn_model_clone.restype = vp
        L.nnn_model_clone.argtypes = [vp]
        L.nnn_model_shape.argtypes = [vp, C.POINTER(C.c_int32)]
        L.nnn_convert_rnnoise_text.restype = C.c_long
        L.nnn_convert_rnnoise_text.argtypes = [C.c_char_p, sz, vp, sz]
        L.nnn_model_from_rnnoise_text.restype = vp
        L.nnn_model_from_rnnoise_text.argtypes = [C.c_char_p, sz]
        L.nnn_batch_create_grouped.restype = vp
        L.nnn_batch_create_grouped.argtypes = [C.POINTER(vp), C.POINTER(i32), i32, i32]
        L.nnn_batch_create.restype = vp
        L.nnn_batch_create.argtypes = [vp, i32, i32]
        L.nnn_batch_destroy.argtypes = [vp]
        L.nnn_batch_num_streams.argtypes = [vp]
        L.nnn_batch_reset.argtypes = [vp]
        L.nnn_batch_process_device.argtypes = [vp, vp, vp, vp, i32, sz, sz, vp]
        L.nnn_batch_process_host.argtypes = [vp, vp, vp, vp, i32, sz, sz]
        L.nnn_batch_process_pcm_device.argtypes = [vp, vp, vp, vp, i32, C.POINTER(PcmLayout), vp]
        L.nnn_batch_process_pcm_host.argtypes = [vp, vp, vp, vp, i32, C.POINTER(PcmLayout)]
        L.nnn_batch_synchronize.argtypes = [vp]
        L.nnn_host_alloc.restype = vp
        L.nnn_host_alloc.argtypes = [sz]
        L.nnn_host_free.restype = None
        L.nnn_host_free.argtypes = [vp]
        L.nnn_tap_info.argtypes = [i32, C.POINTER(i32), C.POINTER(i32)]
        L.nnn_batch_read_tap.argtypes = [vp, i32, vp, sz]
        L.nnn_batch_set_profiling.argtypes = [vp, i32]
        L.nnn_batch_kernel_name.restype = C.c_char_p
        L.nnn_batch_kernel_name.argtypes = [i32]
        L.nnn_batch_read_kernel_times.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_int64), i32]
        L.nnn_batch_set_graph.argtypes = [vp, i32]
        L.nnn_batch_clone.restype = vp
        L.nnn_batch_clone.argtypes = [vp]
        L.nnn_batch_state_bytes.restype = sz
        L.nnn_batch_state_bytes.argtypes = [vp]
        L.nnn_batch_save_state.argtypes = [vp, vp, sz]
        L.nnn_batch_load_state.argtypes = [vp, vp, sz]
        L.nnn_batch_set_taps.argtypes = [vp, i32]
        L.nnn_batch_set_schedule.argtypes = [vp, i32, i32]
        L.nnn_debug_activations.argtypes = [i32, i32, vp, vp, i32]
        L.nnn_batch_set_pipeline.argtypes = [vp, i32]
        L.nnn_batch_set_inputs_ready.argtypes = [vp, i32]
        for name, at in (("nnn_batch_set_frame_log", [vp, vp, sz]), ("nnn_batch_fault", [vp]), ("nnn_batch_debug_withhold_flag", [vp, i32]),
                         ("nnn_batch_create_opts", [C.POINTER(vp), C.POINTER(i32), i32, i32, C.POINTER(BatchOpts)]),
                         ("nnn_batch_max_group_frames", [vp]), ("nnn_batch_device_bytes", [vp]), ("nnn_batch_set_back_end", [vp, i32])):
            if hasattr(L, name):   # (experimental builds of older sources, loaded through NNN_LIBRARY, lack the newest entry points)
                getattr(L, name).argtypes = at
        if hasattr(L, "nnn_batch_create_opts"):
            L.nnn_batch_create_opts.restype = vp
            L.nnn_batch_device_bytes.restype = sz
        if hasattr(L, "nnn_node_create"):
            L.nnn_node_create.restype = vp
            L.nnn_node_create.argtypes = [vp, i32, C.POINTER(i32), i32, C.POINTER(BatchOpts)]
            L.nnn_node_destroy.argtypes = [vp]
            L.nnn_node_num_streams.argtypes = [vp]
            L.nnn_node_num_shards.argtypes = [vp]
            L.nnn_node_shard.argtypes = [vp, i32, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)]
            L.nnn_node_batch.restype = vp
            L.nnn_node_batch.argtypes = [vp, i32]
            L.nnn_node_reset.argtypes = [vp]
            L.nnn_node_process_host.argtypes = [vp, vp, vp, vp, i32, sz, sz]
            L.nnn_node_process_pcm_host.argtypes = [vp, vp, vp, vp, i32, C.POINTER(PcmLayout)]
            L.nnn_node_process_device.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), i32, sz, sz]
            L.nnn_node_process_device_streams.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), i32, i32, sz, sz]
            L.nnn_node_shard_cpus.restype = C.c_char_p
            L.nnn_node_shard_cpus.argtypes = [vp, i32]
            L.nnn_node_synchronize.argtypes = [vp]
            L.nnn_node_fault.argtypes = [vp]
        L.nnn_train_create.restype = vp
        L.nnn_train_create.argtypes = [i32, i32]
        L.nnn_train_destroy.argtypes = [vp]
        L.nnn_train_reset.argtypes = [vp]
        L.nnn_train_process_device.argtypes = [vp, vp, vp, vp, vp, vp, vp, i32, sz, sz, vp]
        L.nnn_train_process_host.argtypes = [vp, vp, vp, vp, vp, vp, vp, i32]
        L.nnn_resampler_create.restype = vp
        L.nnn_resampler_create.argtypes = [i32, C.c_double, i32]
        L.nnn_resampler_destroy.argtypes = [vp]
        L.nnn_resampler_reset.argtypes = [vp]
        L.nnn_resampler_max_output.restype = C.c_long
        L.nnn_resampler_max_output.argtypes = [vp, C.c_long]
        L.nnn_resampler_process_device.argtypes = [vp, vp, C.c_long, sz, vp, C.c_long, sz, C.POINTER(C.c_long), vp]
        L.nnn_resampler_process_host.argtypes = [vp, vp, C.c_long, vp, C.c_long, C.POINTER(C.c_long)]
        L.nnn_last_error.restype = C.c_char_p
        L.rnnoise_create.restype = vp
        L.rnnoise_create.argtypes = [vp]
        L.rnnoise_destroy.argtypes = [vp]
        L.rnnoise_process_frame.restype = C.c_float
        L.rnnoise_process_frame.argtypes = [vp, vp, vp]
        L.rnnoise_model_free.argtypes = [vp]

    def error(self):
        return self.L.nnn_last_error().decode()

    def check(self, rc):
        if rc != 0:
            raise RuntimeError("nnnoiseless_amd: " + self.error())


def ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def as_f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)
