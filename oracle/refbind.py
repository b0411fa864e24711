"""TEST INFRASTRUCTURE — ctypes binding of oracle/_ref/libsphref.so (the reference's own kernels,
compiled by oracle/ref/Makefile from /root/reference where it lies).

Only importable where oracle/_ref/ has been built (the build container). Used by tests/ and by
tests/golden/make_golden.py to pin oracle/sph_oracle.c; never by the product path.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "_ref", "libsphref.so")

STAGES = ["clearBuffers", "hashParticles", "sort", "sortPostPass", "indexx", "indexPostPass", "findNeighbors",
          "computeDensity", "computeForcesAndInitPressure", "computeElasticForces", "predictPositions",
          "predictDensity", "correctPressure", "computePressureForceAcceleration", "integrate",
          "clearMembraneBuffers", "computeInteractionWithMembranes", "computeInteractionWithMembranes_finalize"]
STAGE_ID = {n: i for i, n in enumerate(STAGES)}


class RefConstants(C.Structure):
    _fields_ = [(n, C.c_float) for n in
                ["rho0", "mass", "timeStep", "simulationScale", "h", "hashGridCellSize", "hashGridCellSizeInv",
                 "simulationScaleInv", "r0", "stiffness", "viscosity", "damping", "gravity_x", "gravity_y",
                 "gravity_z", "delta", "xmin", "xmax", "ymin", "ymax", "zmin", "zmax"]] + \
               [(n, C.c_int) for n in ["gridCellsX", "gridCellsY", "gridCellsZ", "gridCellCount", "maxIteration"]] + \
               [(n, C.c_double) for n in ["beta", "Wpoly6Coefficient", "gradWspikyCoefficient",
                                          "del2WviscosityCoefficient"]]


class RefScene(C.Structure):
    _fields_ = [("N", C.c_int), ("numOfLiquidP", C.c_int), ("numOfElasticP", C.c_int), ("numOfBoundaryP", C.c_int),
                ("numOfMembranes", C.c_int), ("position", C.POINTER(C.c_float)), ("velocity", C.POINTER(C.c_float)),
                ("elasticConnections", C.POINTER(C.c_float)), ("membraneData", C.POINTER(C.c_int)),
                ("particleMembranesList", C.POINTER(C.c_int))]


_BUF_DTYPE = {"position": np.float32, "velocity": np.float32, "sortedPosition": np.float32,
              "sortedVelocity": np.float32, "acceleration": np.float32, "neighborMap": np.float32,
              "particleIndex": np.uint32, "particleIndexBack": np.uint32, "gridCellIndex": np.uint32,
              "gridCellIndexFixedUp": np.uint32, "pressure": np.float32, "rho": np.float32}


def available():
    return os.path.exists(LIB_PATH)


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(LIB_PATH)
        L.ref_create.restype = C.c_void_p
        L.ref_create.argtypes = [C.c_int, C.c_float, C.c_float, C.c_float, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                 C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
        L.ref_destroy.argtypes = [C.c_void_p]
        L.ref_run.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.ref_step.argtypes = [C.c_void_p, C.c_int]
        L.ref_update_muscles.argtypes = [C.c_void_p, C.c_void_p]
        L.ref_buffer.restype = C.c_size_t
        L.ref_buffer.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_void_p)]
        L.ref_get_constants.argtypes = [C.POINTER(RefConstants)]
        L.ref_generate_scene.argtypes = [C.POINTER(RefScene)]
        L.ref_free_scene.argtypes = [C.POINTER(RefScene)]
        L.ref_load_configuration.argtypes = [C.c_char_p, C.POINTER(RefScene)]
        L.ref_save_configuration.argtypes = [C.c_char_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
        _lib = L
    return _lib


def constants():
    c = RefConstants()
    lib().ref_get_constants(C.byref(c))
    return c


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class RefSolver:
    """Reference kernels on the host, one method per owOpenCLSolver::_run* stage."""

    def __init__(self, position, velocity, box, grid, elastic=None, membranes=None, particle_membranes=None,
                 elastic_offset=0, threads=8):
        self.N = position.shape[0]
        self._keep = [np.ascontiguousarray(position, np.float32), np.ascontiguousarray(velocity, np.float32)]
        ne = 0 if elastic is None else elastic.shape[0] // 32
        nm = 0 if membranes is None else membranes.size // 3
        el = None if elastic is None else np.ascontiguousarray(elastic, np.float32)
        mb = None if membranes is None else np.ascontiguousarray(membranes, np.int32)
        pm = None if particle_membranes is None else np.ascontiguousarray(particle_membranes, np.int32)
        self._keep += [el, mb, pm]
        self.h = lib().ref_create(self.N, box[0], box[1], box[2], grid[0], grid[1], grid[2], _ptr(self._keep[0]),
                                  _ptr(self._keep[1]), ne, elastic_offset, _ptr(el), nm, _ptr(mb), _ptr(pm), threads)

    def close(self):
        if self.h:
            lib().ref_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def run(self, stage, iteration=0):
        return lib().ref_run(self.h, STAGE_ID[stage], iteration)

    def step(self, iteration=0):
        return lib().ref_step(self.h, iteration)

    def update_muscles(self, signal):
        s = np.ascontiguousarray(signal, np.float32)
        assert s.size == 100
        lib().ref_update_muscles(self.h, _ptr(s))

    def buffer(self, name):
        """Copy of a reference buffer in the reference's own layout (SURVEY table 2.2)."""
        p = C.c_void_p()
        nbytes = lib().ref_buffer(self.h, name.encode(), C.byref(p))
        if nbytes == 0:
            raise KeyError(name)
        dt = np.dtype(_BUF_DTYPE[name])
        arr = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(nbytes,)).view(dt).copy()
        return arr


def load_configuration(directory):
    """The reference's own owHelper::preLoadConfiguration + loadConfiguration on <directory>/configuration/{position,velocity,
    elasticconnections}.txt (owHelper.cpp:1431-1545)."""
    s = RefScene()
    rc = lib().ref_load_configuration(directory.encode(), C.byref(s))
    if rc:
        raise RuntimeError("ref_load_configuration failed: %d" % rc)
    N, ne = s.N, s.numOfElasticP
    out = dict(N=N, numOfLiquidP=s.numOfLiquidP, numOfElasticP=ne, numOfBoundaryP=s.numOfBoundaryP,
               position=np.ctypeslib.as_array(s.position, shape=(N, 4)).copy(),
               velocity=np.ctypeslib.as_array(s.velocity, shape=(N, 4)).copy(),
               elastic=np.ctypeslib.as_array(s.elasticConnections, shape=(ne * 32, 4)).copy() if ne else None)
    lib().ref_free_scene(C.byref(s))
    return out


def save_configuration(directory, position, connections, membranes4, n_elastic, n_liquid, first):
    """The reference's own owHelper::loadConfigurationToFile (owHelper.cpp:1640-1672) into <directory>/buffers/.
    membranes4: int32 [M, 4] (the reference indexes its membrane array with stride 4)."""
    pos = np.ascontiguousarray(position, np.float32)
    con = None if connections is None else np.ascontiguousarray(connections, np.float32)
    mem = None if membranes4 is None else np.ascontiguousarray(membranes4, np.int32)
    rc = lib().ref_save_configuration(directory.encode(), pos.ctypes.data, pos.shape[0], None if con is None else con.ctypes.data,
                                      None if mem is None else mem.ctypes.data, n_elastic, n_liquid,
                                      0 if mem is None else mem.shape[0], int(first))
    if rc:
        raise RuntimeError("ref_save_configuration failed: %d" % rc)


def generate_worm_scene():
    """owHelper::generateConfiguration output (BASELINE config #3)."""
    s = RefScene()
    lib().ref_generate_scene(C.byref(s))
    N, ne, nm = s.N, s.numOfElasticP, s.numOfMembranes
    out = dict(
        N=N, numOfLiquidP=s.numOfLiquidP, numOfElasticP=ne, numOfBoundaryP=s.numOfBoundaryP, numOfMembranes=nm,
        position=np.ctypeslib.as_array(s.position, shape=(N, 4)).copy(),
        velocity=np.ctypeslib.as_array(s.velocity, shape=(N, 4)).copy(),
        elastic=np.ctypeslib.as_array(s.elasticConnections, shape=(ne * 32, 4)).copy() if ne else None,
        membranes=np.ctypeslib.as_array(s.membraneData, shape=(nm, 3)).copy() if nm else None,
        particle_membranes=np.ctypeslib.as_array(s.particleMembranesList, shape=(ne, 7)).copy() if ne else None)
    lib().ref_free_scene(C.byref(s))
    return out
