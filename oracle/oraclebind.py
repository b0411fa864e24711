"""TEST INFRASTRUCTURE — ctypes binding of oracle/libsphoracle.so (the CPU restatement, sph_oracle.c).

May be imported only by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libsphoracle.so")

STAGES = ["clearBuffers", "hashParticles", "sort", "sortPostPass", "indexx", "indexPostPass", "findNeighbors",
          "computeDensity", "computeForcesAndInitPressure", "computeElasticForces", "predictPositions",
          "predictDensity", "correctPressure", "computePressureForceAcceleration", "integrate",
          "clearMembraneBuffers", "computeInteractionWithMembranes", "computeInteractionWithMembranes_finalize"]
STAGE_ID = {n: i for i, n in enumerate(STAGES)}


class OracleParams(C.Structure):
    _fields_ = [("N", C.c_int32), ("gridCellsX", C.c_int32), ("gridCellsY", C.c_int32), ("gridCellsZ", C.c_int32),
                ("gridCellCount", C.c_int32), ("cellIdMask", C.c_uint32)] + \
               [(n, C.c_float) for n in
                ["h", "hashGridCellSize", "hashGridCellSizeInv", "simulationScale", "simulationScaleInv",
                 "xmin", "xmax", "ymin", "ymax", "zmin", "zmax", "r0", "mass", "rho0", "timeStep", "viscosity",
                 "delta", "gravity_x", "gravity_y", "gravity_z", "surfTensCoeff"]] + \
               [(n, C.c_double) for n in ["Wpoly6Coefficient", "gradWspikyCoefficient", "del2WviscosityCoefficient"]] + \
               [(n, C.c_int32) for n in ["numOfElasticP", "elasticOffset", "muscleCount", "numOfMembranes",
                                         "maxIteration", "threads"]]


_BUF_DTYPE = {"position": np.float32, "velocity": np.float32, "sortedPosition": np.float32,
              "sortedVelocity": np.float32, "acceleration": np.float32, "neighborMap": np.float32,
              "neighborIds": np.int32, "particleIndex": np.uint32, "particleIndexBack": np.uint32,
              "gridCellIndex": np.uint32, "gridCellIndexFixedUp": np.uint32, "pressure": np.float32,
              "rho": np.float32}


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE, "libsphoracle.so"])


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        L = C.CDLL(LIB_PATH)
        L.sph_oracle_create.restype = C.c_void_p
        L.sph_oracle_create.argtypes = [C.POINTER(OracleParams)] + [C.c_void_p] * 5
        L.sph_oracle_destroy.argtypes = [C.c_void_p]
        L.sph_oracle_run.argtypes = [C.c_void_p, C.c_int]
        L.sph_oracle_step.argtypes = [C.c_void_p]
        L.sph_oracle_update_muscles.argtypes = [C.c_void_p, C.c_void_p]
        L.sph_oracle_buffer.restype = C.c_size_t
        L.sph_oracle_buffer.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_void_p)]
        L.sph_oracle_stage_seconds.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
        L.sph_oracle_surf_tens_coeff.restype = C.c_float
        L.sph_oracle_surf_tens_coeff.argtypes = [C.c_double, C.c_float, C.c_float]
        _lib = L
    return _lib


def make_params(cfg, threads=1):
    """cfg: dict with the keys of include/sphmi.h's sph_config (see sphmi.config_dict)."""
    p = OracleParams()
    for name, _ in OracleParams._fields_:
        if name == "threads":
            p.threads = threads
        elif name == "surfTensCoeff":
            p.surfTensCoeff = cfg.get("surfTensCoeff", lib().sph_oracle_surf_tens_coeff(
                cfg["Wpoly6Coefficient"], cfg["h"], cfg["simulationScale"]))
        else:
            setattr(p, name, cfg[name])
    return p


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class OracleSolver:
    def __init__(self, cfg, position, velocity, elastic=None, membranes=None, particle_membranes=None, threads=1):
        self.params = make_params(cfg, threads)
        self.N = self.params.N
        pos = np.ascontiguousarray(position, np.float32)
        vel = np.ascontiguousarray(velocity, np.float32)
        assert pos.size == 4 * self.N and vel.size == 4 * self.N
        el = None if elastic is None else np.ascontiguousarray(elastic, np.float32)
        mb = None if membranes is None else np.ascontiguousarray(membranes, np.int32)
        pm = None if particle_membranes is None else np.ascontiguousarray(particle_membranes, np.int32)
        self.h = lib().sph_oracle_create(C.byref(self.params), _ptr(pos), _ptr(vel), _ptr(el), _ptr(mb), _ptr(pm))

    def close(self):
        if getattr(self, "h", None):
            lib().sph_oracle_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def run(self, stage):
        return lib().sph_oracle_run(self.h, STAGE_ID[stage])

    def step(self):
        return lib().sph_oracle_step(self.h)

    def update_muscles(self, signal):
        s = np.ascontiguousarray(signal, np.float32)
        assert s.size == self.params.muscleCount
        lib().sph_oracle_update_muscles(self.h, _ptr(s))

    def buffer(self, name):
        p = C.c_void_p()
        nbytes = lib().sph_oracle_buffer(self.h, name.encode(), C.byref(p))
        if nbytes == 0:
            raise KeyError(name)
        dt = np.dtype(_BUF_DTYPE[name])
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(nbytes,)).view(dt).copy()

    def stage_seconds(self):
        out = (C.c_double * len(STAGES))()
        lib().sph_oracle_stage_seconds(self.h, out)
        return dict(zip(STAGES, list(out)))
