"""sphmi — ctypes binding of libsphmi.so / libsphmi_host.so (include/sphmi.h, include/sphmi_host.h).

`owHIPSolver` mirrors the reference's `owOpenCLSolver` (src/owOpenCLSolver.h:28-62) method for method, and
`owPhysicsFluidSimulator` mirrors the stage order of `simulationStep()` (src/owPhysicsFluidSimulator.cpp:79-149),
so tests read like calls into the reference. There is no CPU fallback: if libsphmi.so or a GPU is missing the
constructor raises.
"""
import ctypes as C
import os

import numpy as np

_PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.environ.get("SPHMI_LIB", os.path.join(_PKG, "libsphmi.so"))  # SPHMI_LIB: A/B builds of the same ABI
HOST_LIB_PATH = os.path.join(_PKG, "libsphmi_host.so")

ABI_VERSION = 2
MAX_NEIGHBOR_COUNT = 32
LIQUID_PARTICLE, ELASTIC_PARTICLE, BOUNDARY_PARTICLE = 1, 2, 3

STAGE_NAMES = ["hash", "sort", "sort_post", "index", "find_neighbors", "density", "forces", "elastic",
               "predict_density", "pressure_force", "integrate", "membranes"]


class SphConfig(C.Structure):
    _fields_ = [("abi_version", C.c_int32), ("particleCount", C.c_int32), ("capacity", C.c_int32),
                ("gridCellsX", C.c_int32), ("gridCellsY", C.c_int32), ("gridCellsZ", C.c_int32),
                ("gridCellCount", C.c_int32), ("cellIdMask", C.c_uint32)] + \
               [(n, C.c_float) for n in
                ["h", "hashGridCellSize", "hashGridCellSizeInv", "simulationScale", "simulationScaleInv",
                 "xmin", "xmax", "ymin", "ymax", "zmin", "zmax", "r0", "mass", "rho0", "timeStep", "viscosity",
                 "delta", "gravity_x", "gravity_y", "gravity_z", "surfTensCoeff"]] + \
               [(n, C.c_double) for n in ["Wpoly6Coefficient", "gradWspikyCoefficient", "del2WviscosityCoefficient"]] + \
               [(n, C.c_int32) for n in ["numOfElasticP", "elasticOffset", "muscleCount", "numOfMembranes",
                                         "maxIteration", "device"]] + \
               [("stream", C.c_void_p)]


class SphSlab(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ["layerLo", "layerHi", "ghostLayers", "hasLower", "hasUpper", "globalIdBits"]]


SLAB_RECORD_WORDS = 9
SLAB_COMPACT_WORDS = 7  # x, y, z, vx, vy, vz, global id (sph_slab_set_record_format)


class SphError(RuntimeError):
    pass


_host = None
_dev = None


def host_lib():
    global _host
    if _host is None:
        if not os.path.exists(HOST_LIB_PATH):
            raise SphError("libsphmi_host.so not built (run `python -c 'import __graft_entry__ as g; g.build()'`)")
        L = C.CDLL(HOST_LIB_PATH)
        L.sphmi_default_config.argtypes = [C.POINTER(SphConfig)]
        L.sphmi_config_set_box.argtypes = [C.POINTER(SphConfig), C.c_double, C.c_double, C.c_double, C.c_uint32]
        L.sphmi_count_particles.argtypes = [C.c_char_p]
        L.sphmi_load_configuration.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_void_p, C.c_void_p,
                                               C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.sphmi_load_elastic_connections.argtypes = [C.c_char_p, C.c_int, C.c_void_p]
        L.sphmi_box_counts.argtypes = [C.POINTER(SphConfig), C.c_double, C.c_double, C.c_double, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int),
                                       C.POINTER(C.c_int)]
        L.sphmi_generate_box.argtypes = [C.POINTER(SphConfig), C.c_double, C.c_double, C.c_double, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float,
                                         C.c_float, C.c_float, C.c_float, C.c_uint64, C.c_void_p, C.c_void_p]
        _box = [C.POINTER(SphConfig), C.c_double, C.c_double, C.c_double, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float,
                C.c_float, C.c_float, C.c_float, C.c_uint64]
        L.sphmi_box_layer_histogram.argtypes = _box + [C.c_void_p, C.c_int]
        L.sphmi_generate_box_slice.argtypes = _box + [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int)]
        L.sphmi_muscle_signal.argtypes = [C.c_int, C.c_void_p, C.c_int]
        L.sphmi_trajectory_info.argtypes = [C.c_char_p] + [C.POINTER(C.c_int)] * 4
        L.sphmi_trajectory_frame.argtypes = [C.c_char_p, C.c_int, C.c_void_p]
        L.sphmi_trajectory_connections.argtypes = [C.c_char_p, C.c_int, C.c_void_p]
        L.sphmi_trajectory_membranes.argtypes = [C.c_char_p, C.c_int, C.c_void_p]
        L.sphmi_worm_counts.argtypes = [C.POINTER(SphConfig), C.c_double, C.c_double, C.c_double] + [C.POINTER(C.c_int)] * 4
        L.sphmi_generate_worm.argtypes = [C.POINTER(SphConfig), C.c_double, C.c_double, C.c_double] + [C.c_void_p] * 5
        L.sphmi_save_configuration.argtypes = [C.c_char_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                               C.c_int, C.c_int]
        _host = L
    return _host


_STAGE_FUNCS = ["sph_run_clear_buffers", "sph_run_hash_particles", "sph_run_sort", "sph_run_sort_post_pass",
                "sph_run_indexx", "sph_run_index_post_pass", "sph_run_find_neighbors",
                "sph_run_pcisph_compute_density", "sph_run_pcisph_compute_forces_and_init_pressure",
                "sph_run_pcisph_compute_elastic_forces", "sph_run_pcisph_predict_positions",
                "sph_run_pcisph_predict_density", "sph_run_pcisph_correct_pressure",
                "sph_run_pcisph_compute_pressure_force_acceleration", "sph_run_clear_membrane_buffers",
                "sph_run_compute_interaction_with_membranes", "sph_run_compute_interaction_with_membranes_finalize"]
EXPORTED_SYMBOLS = ["sph_create", "sph_destroy", "sph_run_pcisph_integrate", "sph_step", "sph_update_muscles",
                    "sph_read_position", "sph_read_velocity", "sph_read_density", "sph_read_particle_index",
                    "sph_read_position_async", "sph_read_position_wait", "sph_host_unregister", "sph_build_info",
                    "sph_read_buffer", "sph_read_neighbor_rows", "sph_synchronize", "sph_set_stage_timing", "sph_get_stage_times",
                    "sph_reset_stage_times", "sph_step_sort_passes", "sph_last_error", "sph_abi_version", "sph_slab_init", "sph_slab_pack", "sph_slab_pack_framed", "sph_slab_step_begin", "sph_slab_step_messages",
                    "sph_slab_rebuild", "sph_particle_count", "sph_slab_read", "sph_slab_rebuild_framed", "sph_slab_rebuild_finish",
                    "sph_slab_liquid_signature", "sph_slab_set_record_format", "sph_stream_wait_event"] + _STAGE_FUNCS
HOST_EXPORTED_SYMBOLS = ["sphmi_default_config", "sphmi_config_set_box", "sphmi_count_particles",
                         "sphmi_load_configuration", "sphmi_load_elastic_connections", "sphmi_box_counts",
                         "sphmi_generate_box", "sphmi_box_layer_histogram", "sphmi_generate_box_slice", "sphmi_muscle_signal", "sphmi_save_configuration", "sphmi_worm_counts",
                         "sphmi_generate_worm", "sphmi_trajectory_info", "sphmi_trajectory_frame", "sphmi_trajectory_connections",
                         "sphmi_trajectory_membranes"]


def hip_runtimes_loaded():
    """Paths of every libamdhip64 mapped into this process (two of them do not both see the GPU)."""
    paths = set()
    try:
        with open("/proc/self/maps") as f:
            for line in f:
                if "libamdhip64" in line:
                    paths.add(line.split()[-1])
    except OSError:
        pass
    return sorted(paths)


def _bind_one_hip_runtime():
    """PyTorch-ROCm ships its own libamdhip64 (same SONAME as /opt/rocm's). Whichever copy is mapped first serves every
    later user, so if none is loaded yet and torch is installed, map torch's copy now — by path, without importing torch —
    and libsphmi.so, a later `import torch` and RCCL all share it, whatever the import order."""
    if hip_runtimes_loaded():
        return
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec and spec.submodule_search_locations:
        cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
        if os.path.exists(cand):
            C.CDLL(cand, mode=C.RTLD_GLOBAL)


def device_lib():
    """Load libsphmi.so (the HIP solver). Raises if it has not been built — there is no fallback."""
    global _dev
    if _dev is None:
        if not os.path.exists(LIB_PATH):
            raise SphError("libsphmi.so not built: the HIP extension is required (no CPU fallback exists)")
        _bind_one_hip_runtime()
        L = C.CDLL(LIB_PATH)
        L.sph_create.argtypes = [C.POINTER(SphConfig), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                 C.POINTER(C.c_void_p)]
        L.sph_destroy.argtypes = [C.c_void_p]
        for f in _STAGE_FUNCS + ["sph_synchronize", "sph_reset_stage_times", "sph_step_sort_passes"]:
            getattr(L, f).argtypes = [C.c_void_p]
        L.sph_run_pcisph_integrate.argtypes = [C.c_void_p, C.c_int]
        L.sph_step.argtypes = [C.c_void_p, C.c_int]
        L.sph_update_muscles.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        for f in ["sph_read_position", "sph_read_velocity", "sph_read_density", "sph_read_particle_index", "sph_read_position_async",
                  "sph_host_unregister"]:
            getattr(L, f).argtypes = [C.c_void_p, C.c_void_p]
        L.sph_read_position_wait.argtypes = [C.c_void_p]
        L.sph_build_info.restype = C.c_char_p
        L.sph_read_buffer.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
        L.sph_read_neighbor_rows.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]
        L.sph_set_stage_timing.argtypes = [C.c_void_p, C.c_int]
        L.sph_get_stage_times.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.c_int]
        L.sph_last_error.restype = C.c_char_p
        L.sph_slab_init.argtypes = [C.c_void_p, C.POINTER(SphSlab), C.c_void_p]
        L.sph_slab_pack.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(C.c_int32)]
        L.sph_slab_pack_framed.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(C.c_int32)]
        L.sph_slab_step_begin.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int32]
        L.sph_slab_step_messages.argtypes = [C.c_void_p, C.POINTER(C.c_int32)]
        L.sph_slab_rebuild.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32]
        L.sph_slab_rebuild_framed.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32]
        L.sph_slab_rebuild_finish.argtypes = [C.c_void_p, C.POINTER(C.c_int32)]
        L.sph_slab_liquid_signature.argtypes = [C.c_void_p, C.POINTER(C.c_uint32)]
        L.sph_slab_set_record_format.argtypes = [C.c_void_p, C.c_int32, C.c_uint32]
        L.sph_stream_wait_event.argtypes = [C.c_void_p, C.c_void_p]
        L.sph_particle_count.argtypes = [C.c_void_p]
        L.sph_slab_read.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        _dev = L
    return _dev


# ----------------------------------------------------------------------------- host helpers
def default_config():
    cfg = SphConfig()
    rc = host_lib().sphmi_default_config(C.byref(cfg))
    if rc:
        raise SphError("sphmi_default_config failed: %d" % rc)
    cfg._box_in_h = (30.0, 20.0, 250.0)  # owPhysicsConstant.h:32-37
    return cfg


def set_box(cfg, xmax_in_h, ymax_in_h, zmax_in_h, cell_id_mask=0xffff):
    rc = host_lib().sphmi_config_set_box(C.byref(cfg), xmax_in_h, ymax_in_h, zmax_in_h, cell_id_mask)
    if rc:
        raise SphError("sphmi_config_set_box failed: %d" % rc)
    cfg._box_in_h = (float(xmax_in_h), float(ymax_in_h), float(zmax_in_h))
    return cfg


def config_dict(cfg):
    """sph_config as a dict with the oracle's key names (oracle/oraclebind.make_params)."""
    d = {n: getattr(cfg, n) for n, _ in SphConfig._fields_ if n not in ("stream",)}
    d["N"] = d["particleCount"]
    return d


def load_configuration(position_file, velocity_file):
    """owHelper::preLoadConfiguration + loadConfiguration (owHelper.cpp:1431-1545)."""
    n = host_lib().sphmi_count_particles(position_file.encode())
    if n <= 0:
        raise SphError("cannot read %s" % position_file)
    pos = np.empty((n, 4), np.float32)
    vel = np.empty((n, 4), np.float32)
    nl, ne, nb = C.c_int(), C.c_int(), C.c_int()
    rc = host_lib().sphmi_load_configuration(position_file.encode(), velocity_file.encode(), n, pos.ctypes.data,
                                             vel.ctypes.data, C.byref(nl), C.byref(ne), C.byref(nb))
    if rc:
        raise SphError("sphmi_load_configuration failed: %d" % rc)
    return pos, vel, dict(numOfLiquidP=nl.value, numOfElasticP=ne.value, numOfBoundaryP=nb.value)


def generate_box(cfg, lx, ly, lz, spacing=None, origin=None, jitter=0.0, seed=20261004):
    """Synthetic pure-liquid box of SURVEY §8(d): liquid lattice + reference boundary shell. Sets cfg.particleCount."""
    nl, nb = C.c_int(), C.c_int()
    bx = cfg._box_in_h
    rc = host_lib().sphmi_box_counts(C.byref(cfg), bx[0], bx[1], bx[2], lx, ly, lz, C.byref(nl), C.byref(nb))
    if rc:
        raise SphError("sphmi_box_counts failed: %d" % rc)
    n = nl.value + nb.value
    r0 = np.float32(cfg.r0)
    if spacing is None:
        spacing = np.float32(0.93) * r0
    if origin is None:
        origin = (np.float32(3) * r0,) * 3
    pos = np.empty((n, 4), np.float32)
    vel = np.empty((n, 4), np.float32)
    rc = host_lib().sphmi_generate_box(C.byref(cfg), bx[0], bx[1], bx[2], lx, ly, lz, spacing, origin[0], origin[1], origin[2], jitter,
                                       seed, pos.ctypes.data, vel.ctypes.data)
    if rc:
        raise SphError("sphmi_generate_box failed: %d" % rc)
    cfg.particleCount = n
    return pos, vel, dict(numOfLiquidP=nl.value, numOfElasticP=0, numOfBoundaryP=nb.value)


def _box_args(cfg, lx, ly, lz, spacing, origin, jitter, seed):
    bx = cfg._box_in_h
    r0 = np.float32(cfg.r0)
    if spacing is None:
        spacing = np.float32(0.93) * r0
    if origin is None:
        origin = (np.float32(3) * r0,) * 3
    return [C.byref(cfg), bx[0], bx[1], bx[2], lx, ly, lz, spacing, origin[0], origin[1], origin[2], jitter, seed]


def box_counts(cfg, lx, ly, lz):
    """(numOfLiquidP, numOfBoundaryP) of generate_box without generating anything."""
    nl, nb = C.c_int(), C.c_int()
    bx = cfg._box_in_h
    rc = host_lib().sphmi_box_counts(C.byref(cfg), bx[0], bx[1], bx[2], lx, ly, lz, C.byref(nl), C.byref(nb))
    if rc:
        raise SphError("sphmi_box_counts failed: %d" % rc)
    return nl.value, nb.value


def box_layer_histogram(cfg, lx, ly, lz, spacing=None, origin=None, jitter=0.0, seed=20261004):
    """Particles of generate_box's scene per z cell layer (int64[gridCellsZ]) — what balanced_cuts_hist needs — without
    materialising the scene."""
    hist = np.zeros(cfg.gridCellsZ, np.int64)
    rc = host_lib().sphmi_box_layer_histogram(*_box_args(cfg, lx, ly, lz, spacing, origin, jitter, seed), hist.ctypes.data, hist.size)
    if rc:
        raise SphError("sphmi_box_layer_histogram failed: %d" % rc)
    return hist


def generate_box_slice(cfg, lx, ly, lz, layer_lo, layer_hi, spacing=None, origin=None, jitter=0.0, seed=20261004):
    """The rows of generate_box's scene whose z cell layer lies in [layer_lo, layer_hi), with their global ids (ascending):
    (position, velocity, global_ids). Does not touch cfg.particleCount."""
    args = _box_args(cfg, lx, ly, lz, spacing, origin, jitter, seed) + [int(max(layer_lo, -(1 << 30))), int(min(layer_hi, 1 << 30))]
    n = C.c_int()
    rc = host_lib().sphmi_generate_box_slice(*args, None, None, None, 0, C.byref(n))
    if rc:
        raise SphError("sphmi_generate_box_slice failed: %d" % rc)
    pos, vel, gid = np.empty((n.value, 4), np.float32), np.empty((n.value, 4), np.float32), np.empty(n.value, np.uint32)
    rc = host_lib().sphmi_generate_box_slice(*args, pos.ctypes.data, vel.ctypes.data, gid.ctypes.data, n.value, C.byref(n))
    if rc:
        raise SphError("sphmi_generate_box_slice failed: %d" % rc)
    return pos, vel, gid


def generate_worm(cfg):
    """The reference's generated worm scene (owHelper::generateConfiguration) for cfg's box. Sets
    cfg.particleCount / numOfElasticP / numOfMembranes / elasticOffset and returns a scene dict."""
    bx = cfg._box_in_h
    ne, nl, nb, nm = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    rc = host_lib().sphmi_worm_counts(C.byref(cfg), bx[0], bx[1], bx[2], C.byref(ne), C.byref(nl), C.byref(nb), C.byref(nm))
    if rc:
        raise SphError("sphmi_worm_counts failed: %d" % rc)
    n = ne.value + nl.value + nb.value
    pos = np.empty((n, 4), np.float32)
    vel = np.empty((n, 4), np.float32)
    elastic = np.empty((ne.value * 32, 4), np.float32)
    membranes = np.empty((nm.value, 3), np.int32)
    pml = np.empty((ne.value, 7), np.int32)
    rc = host_lib().sphmi_generate_worm(C.byref(cfg), bx[0], bx[1], bx[2], pos.ctypes.data, vel.ctypes.data, elastic.ctypes.data,
                                        membranes.ctypes.data, pml.ctypes.data)
    if rc:
        raise SphError("sphmi_generate_worm failed: %d" % rc)
    cfg.particleCount, cfg.numOfElasticP, cfg.numOfMembranes, cfg.elasticOffset = n, ne.value, nm.value, 0
    return dict(cfg=cfg, position=pos, velocity=vel, elastic=elastic, membranes=membranes, particle_membranes=pml,
                numOfLiquidP=nl.value, numOfElasticP=ne.value, numOfBoundaryP=nb.value)


def save_configuration(directory, position, num_elastic, num_liquid, connections=None, membranes=None, first=True):
    """owHelper::loadConfigurationToFile (owHelper.cpp:1640-1672): the `-l_to` trajectory dump."""
    pos = np.ascontiguousarray(position, np.float32)
    con = None if connections is None else np.ascontiguousarray(connections, np.float32)
    mem = None if membranes is None else np.ascontiguousarray(membranes, np.int32)
    rc = host_lib().sphmi_save_configuration(directory.encode(), pos.ctypes.data, pos.shape[0], num_elastic, num_liquid,
                                             None if con is None else con.ctypes.data,
                                             None if mem is None else mem.ctypes.data,
                                             0 if mem is None else mem.shape[0], int(first))
    if rc:
        raise SphError("sphmi_save_configuration failed: %d" % rc)


def load_trajectory(directory):
    """owHelper::loadConfigurationFromFile (owHelper.cpp:1674-1741): read back a `-l_to` dump. Returns a dict with
    numOfElasticP, numOfLiquidP, frames[F, P, 4] and, when present, connections[32*E, 4] and membranes[M, 4]."""
    L = host_lib()
    ne, nl, nf, nm = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    d = directory.encode()
    if L.sphmi_trajectory_info(d, C.byref(ne), C.byref(nl), C.byref(nf), C.byref(nm)):
        raise SphError("no trajectory in %s" % directory)
    P = ne.value + nl.value
    frames = np.empty((nf.value, P, 4), np.float32)
    for f in range(nf.value):
        if L.sphmi_trajectory_frame(d, f, frames[f].ctypes.data):
            raise SphError("trajectory frame %d unreadable" % f)
    out = dict(numOfElasticP=ne.value, numOfLiquidP=nl.value, frames=frames, connections=None, membranes=None)
    if ne.value:
        con = np.empty((32 * ne.value, 4), np.float32)
        if L.sphmi_trajectory_connections(d, ne.value, con.ctypes.data) == 0:
            out["connections"] = con
    if nm.value:
        mem = np.empty((nm.value, 4), np.int32)
        if L.sphmi_trajectory_membranes(d, nm.value, mem.ctypes.data) == 0:
            out["membranes"] = mem
    return out


def muscle_signal(step, muscle_count=100):
    out = np.zeros(muscle_count, np.float32)
    rc = host_lib().sphmi_muscle_signal(step, out.ctypes.data, muscle_count)
    if rc:
        raise SphError("sphmi_muscle_signal failed: %d" % rc)
    return out


# ----------------------------------------------------------------------------- device solver
_BUF_DTYPE = {"position": np.float32, "velocity": np.float32, "sortedPosition": np.float32,
              "sortedVelocity": np.float32, "acceleration": np.float32, "neighborMap": np.float32,
              "neighborIds": np.int32, "particleIndex": np.uint32, "particleIndexBack": np.uint32,
              "gridCellIndex": np.uint32, "gridCellIndexFixedUp": np.uint32, "pressure": np.float32,
              "rho": np.float32, "debugCounters": np.uint32, "diagnosticTrace": np.uint32}


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def build_info():
    """sph_build_info() of the loaded libsphmi.so ("DIAG" in it: a timing-only variant with invalid results)."""
    return device_lib().sph_build_info().decode()


class owHIPSolver:
    """Counterpart of owOpenCLSolver (src/owOpenCLSolver.h:28-62): same methods, backed by libsphmi.so."""

    def __init__(self, cfg, position_cpp, velocity_cpp, elasticConnectionsData_cpp=None, membraneData_cpp=None,
                 particleMembranesList_cpp=None):
        self._L = device_lib()
        self.cfg = cfg
        self.N = cfg.particleCount
        pos = np.ascontiguousarray(position_cpp, np.float32)
        vel = np.ascontiguousarray(velocity_cpp, np.float32)
        if pos.size != 4 * self.N or vel.size != 4 * self.N:
            raise SphError("position/velocity must hold 4*particleCount floats")
        el = None if elasticConnectionsData_cpp is None else np.ascontiguousarray(elasticConnectionsData_cpp, np.float32)
        mb = None if membraneData_cpp is None else np.ascontiguousarray(membraneData_cpp, np.int32)
        pm = None if particleMembranesList_cpp is None else np.ascontiguousarray(particleMembranesList_cpp, np.int32)
        h = C.c_void_p()
        self._chk(self._L.sph_create(C.byref(cfg), _ptr(pos), _ptr(vel), _ptr(el), _ptr(mb), _ptr(pm), C.byref(h)))
        self._h = h

    def _chk(self, rc):
        if rc != 0:
            raise SphError("libsphmi: %s (status %d)" % (self._L.sph_last_error().decode(), rc))
        return 0

    def close(self):
        if getattr(self, "_h", None):
            self._L.sph_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # --- owOpenCLSolver::_run* (owOpenCLSolver.h:37-56) ---
    def _runClearBuffers(self): return self._chk(self._L.sph_run_clear_buffers(self._h))
    def _runHashParticles(self): return self._chk(self._L.sph_run_hash_particles(self._h))
    def _runSort(self): return self._chk(self._L.sph_run_sort(self._h))
    def _runSortPostPass(self): return self._chk(self._L.sph_run_sort_post_pass(self._h))
    def _runIndexx(self): return self._chk(self._L.sph_run_indexx(self._h))
    def _runIndexPostPass(self): return self._chk(self._L.sph_run_index_post_pass(self._h))
    def _runFindNeighbors(self): return self._chk(self._L.sph_run_find_neighbors(self._h))
    def _run_pcisph_computeDensity(self): return self._chk(self._L.sph_run_pcisph_compute_density(self._h))
    def _run_pcisph_computeForcesAndInitPressure(self): return self._chk(self._L.sph_run_pcisph_compute_forces_and_init_pressure(self._h))
    def _run_pcisph_computeElasticForces(self): return self._chk(self._L.sph_run_pcisph_compute_elastic_forces(self._h))
    def _run_pcisph_predictPositions(self): return self._chk(self._L.sph_run_pcisph_predict_positions(self._h))
    def _run_pcisph_predictDensity(self): return self._chk(self._L.sph_run_pcisph_predict_density(self._h))
    def _run_pcisph_correctPressure(self): return self._chk(self._L.sph_run_pcisph_correct_pressure(self._h))
    def _run_pcisph_computePressureForceAcceleration(self): return self._chk(self._L.sph_run_pcisph_compute_pressure_force_acceleration(self._h))
    def _run_pcisph_integrate(self, iterationCount): return self._chk(self._L.sph_run_pcisph_integrate(self._h, iterationCount))
    def _run_clearMembraneBuffers(self): return self._chk(self._L.sph_run_clear_membrane_buffers(self._h))
    def _run_computeInteractionWithMembranes(self): return self._chk(self._L.sph_run_compute_interaction_with_membranes(self._h))
    def _run_computeInteractionWithMembranes_finalize(self): return self._chk(self._L.sph_run_compute_interaction_with_membranes_finalize(self._h))

    def updateMuscleActivityData(self, signal):
        s = np.ascontiguousarray(signal, np.float32)
        return self._chk(self._L.sph_update_muscles(self._h, _ptr(s), s.size))

    def read_position_buffer(self, out=None):
        out = np.empty((self.N, 4), np.float32) if out is None else out
        self._chk(self._L.sph_read_position(self._h, _ptr(out)))
        return out

    def read_position_buffer_async(self, out):
        """read_position_buffer without the wait (sph_read_position_async): the copy runs under the next step; `out` (C-contiguous
        float32, 4N) is valid after wait_position_buffer(). The array is page-locked in place on first use and kept alive here."""
        if not (isinstance(out, np.ndarray) and out.dtype == np.float32 and out.flags.c_contiguous and out.size == 4 * self.N):
            raise SphError("read_position_buffer_async needs a C-contiguous float32 array of 4*N elements")
        self._async_out = out
        self._chk(self._L.sph_read_position_async(self._h, _ptr(out)))
        return out

    def wait_position_buffer(self):
        return self._chk(self._L.sph_read_position_wait(self._h))

    def read_velocity_buffer(self, out=None):
        out = np.empty((self.N, 4), np.float32) if out is None else out
        self._chk(self._L.sph_read_velocity(self._h, _ptr(out)))
        return out

    def read_density_buffer(self, out=None):
        out = np.empty(self.N, np.float32) if out is None else out
        self._chk(self._L.sph_read_density(self._h, _ptr(out)))
        return out

    def read_particleIndex_buffer(self, out=None):
        out = np.empty((self.N, 2), np.uint32) if out is None else out
        self._chk(self._L.sph_read_particle_index(self._h, _ptr(out)))
        return out

    # --- extras ---
    def step(self, iterationCount=0):
        """Fused fast path == the stage sequence of simulationStep()."""
        return self._chk(self._L.sph_step(self._h, iterationCount))

    def synchronize(self):
        return self._chk(self._L.sph_synchronize(self._h))

    def buffer(self, name):
        need = C.c_size_t()
        self._chk(self._L.sph_read_buffer(self._h, name.encode(), None, 0, C.byref(need)))
        out = np.empty(need.value // np.dtype(_BUF_DTYPE[name]).itemsize, _BUF_DTYPE[name])
        self._chk(self._L.sph_read_buffer(self._h, name.encode(), _ptr(out), need.value, None))
        return out

    def neighbor_rows(self, first, count):
        """Neighbour ids and scaled distances of the sorted particles [first, first + count): two (count, 32) arrays."""
        ids = np.empty((count, 32), np.int32)
        dist = np.empty((count, 32), np.float32)
        self._chk(self._L.sph_read_neighbor_rows(self._h, first, count, _ptr(ids), _ptr(dist)))
        return ids, dist

    # --- slab decomposition (include/sphmi.h, "Spatial decomposition") ---
    def slab_init(self, slab, global_ids):
        g = np.ascontiguousarray(global_ids, np.uint32)
        return self._chk(self._L.sph_slab_init(self._h, C.byref(slab), _ptr(g)))

    def slab_pack(self, msg_down_ptr, msg_up_ptr, cap_records):
        counts = (C.c_int32 * 3)()
        self._chk(self._L.sph_slab_pack(self._h, msg_down_ptr, msg_up_ptr, cap_records, counts))
        return counts[0], counts[1], counts[2]

    def slab_pack_framed(self, frame_down_ptr, frame_up_ptr, cap_records):
        counts = (C.c_int32 * 3)()
        self._chk(self._L.sph_slab_pack_framed(self._h, frame_down_ptr, frame_up_ptr, cap_records, counts))
        return counts[0], counts[1], counts[2]

    def slab_step_begin(self, iterationCount, frame_down_ptr, frame_up_ptr, cap_records):
        self._chk(self._L.sph_slab_step_begin(self._h, iterationCount, frame_down_ptr, frame_up_ptr, cap_records))

    def slab_step_messages(self):
        counts = (C.c_int32 * 2)()
        self._chk(self._L.sph_slab_step_messages(self._h, counts))
        return counts[0], counts[1]

    def slab_rebuild(self, recv_down_ptr, n_down, recv_up_ptr, n_up):
        self._chk(self._L.sph_slab_rebuild(self._h, recv_down_ptr, n_down, recv_up_ptr, n_up))
        self.N = self._L.sph_particle_count(self._h)
        return self.N

    def slab_rebuild_framed(self, frame_down_ptr, cap_down_records, frame_up_ptr, cap_up_records):
        """Asynchronous rebuild from complete received frames; the new count arrives with slab_rebuild_finish()."""
        self._chk(self._L.sph_slab_rebuild_framed(self._h, frame_down_ptr, cap_down_records, frame_up_ptr, cap_up_records))

    def slab_rebuild_finish(self):
        """(kept, records from below, records from above, nothing_merged) of the last slab_rebuild_framed; blocks."""
        c = (C.c_int32 * 4)()
        self._chk(self._L.sph_slab_rebuild_finish(self._h, c))
        if not c[3]:
            self.N = self._L.sph_particle_count(self._h)
        return c[0], c[1], c[2], bool(c[3])

    def slab_liquid_signature(self):
        b = C.c_uint32()
        self._chk(self._L.sph_slab_liquid_signature(self._h, C.byref(b)))
        return b.value

    def slab_set_record_format(self, words, type_bits=0):
        return self._chk(self._L.sph_slab_set_record_format(self._h, words, type_bits))

    def stream_wait_event(self, hip_event_handle):
        return self._chk(self._L.sph_stream_wait_event(self._h, C.c_void_p(hip_event_handle)))

    def slab_read(self):
        n = self._L.sph_particle_count(self._h)
        pos, vel = np.empty((n, 4), np.float32), np.empty((n, 4), np.float32)
        gid, owned = np.empty(n, np.uint32), np.empty(n, np.uint32)
        self._chk(self._L.sph_slab_read(self._h, _ptr(pos), _ptr(vel), _ptr(gid), _ptr(owned)))
        return pos, vel, gid, owned

    def set_stage_timing(self, enable=True):
        return self._chk(self._L.sph_set_stage_timing(self._h, int(enable)))

    def reset_stage_times(self):
        return self._chk(self._L.sph_reset_stage_times(self._h))

    def step_sort_passes(self):
        """Radix passes the sort of the fused step takes for this solver (24 algorithmic bytes per particle each)."""
        n = self._L.sph_step_sort_passes(self._h)
        if n < 0:
            self._chk(n)
        return n

    def stage_times(self):
        n = len(STAGE_NAMES)
        ms = (C.c_double * n)()
        cnt = (C.c_int64 * n)()
        self._chk(self._L.sph_get_stage_times(self._h, ms, cnt, n))
        return {STAGE_NAMES[i]: (ms[i], cnt[i]) for i in range(n)}


class owPhysicsFluidSimulator:
    """Stage order of owPhysicsFluidSimulator::simulationStep() (owPhysicsFluidSimulator.cpp:79-149)."""

    def __init__(self, cfg, position_cpp, velocity_cpp, elasticConnectionsData_cpp=None, membraneData_cpp=None,
                 particleMembranesList_cpp=None, fused=True, muscles=False):
        self.ocl_solver = owHIPSolver(cfg, position_cpp, velocity_cpp, elasticConnectionsData_cpp, membraneData_cpp,
                                      particleMembranesList_cpp)
        self.cfg = cfg
        self.iterationCount = 0
        self.fused = fused
        self.muscles = muscles
        self.position_cpp = np.array(position_cpp, np.float32).reshape(-1, 4)

    def simulationStep(self, read_back=True, async_read_back=False):
        s = self.ocl_solver
        if self.fused:
            s.step(self.iterationCount)
        else:
            s._runClearBuffers(); s._runHashParticles(); s._runSort(); s._runSortPostPass(); s._runIndexx()
            s._runIndexPostPass(); s._runFindNeighbors()
            s._run_pcisph_computeDensity(); s._run_pcisph_computeForcesAndInitPressure()
            s._run_pcisph_computeElasticForces()
            it = 0
            while True:
                s._run_pcisph_predictPositions(); s._run_pcisph_predictDensity(); s._run_pcisph_correctPressure()
                s._run_pcisph_computePressureForceAcceleration()
                it += 1
                if it >= self.cfg.maxIteration:
                    break
            s._run_pcisph_integrate(self.iterationCount)
            s._run_clearMembraneBuffers(); s._run_computeInteractionWithMembranes()
            s._run_computeInteractionWithMembranes_finalize()
        if read_back and async_read_back:  # the copy overlaps the next step; getPosition_cpp() waits for it
            s.read_position_buffer_async(self.position_cpp)
        elif read_back:
            s.read_position_buffer(self.position_cpp)
        if self.muscles:  # signals computed after step t drive step t+1 (owPhysicsFluidSimulator.cpp:134-141)
            s.updateMuscleActivityData(muscle_signal(self.iterationCount, self.cfg.muscleCount))
        self.iterationCount += 1

    def getPosition_cpp(self):
        self.ocl_solver.wait_position_buffer()  # (no-op unless simulationStep(async_read_back=True) left a copy in flight)
        return self.position_cpp

    def getDensity_cpp(self): return self.ocl_solver.read_density_buffer()
    def getParticleIndex_cpp(self): return self.ocl_solver.read_particleIndex_buffer()
