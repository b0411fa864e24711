"""Headless counterpart of the reference viewer's per-frame feed.

The GLUT viewer (owWorldSimulation.cpp:100-140, out of scope) pulls three things from the simulator every frame:
`getPosition_cpp()` (orig order), `getDensity_cpp()` (SORTED order) and `getParticleIndex_cpp()` ((cell, orig id) pairs in
sorted order), inverts the permutation itself (`p_indexb[2*p_indexb[2*i+1]] = i`) and colours particle i by
`density[p_indexb[2*i]]`. `frame()` does the same inversion and returns per-particle arrays in orig order; `density_colour()`
is the viewer's blue -> cyan -> green -> yellow -> red ramp over 0..5 % compression; `write_vtk` / `write_npz` replace the GL
window with files ParaView or numpy can open.
"""
import numpy as np


def invert_particle_index(particle_index):
    """orig id -> sorted position, from the (cell, orig id) pairs the solver returns (owWorldSimulation.cpp:112-116)."""
    pi = np.asarray(particle_index, np.uint32).reshape(-1, 2)
    back = np.empty(pi.shape[0], np.uint32)
    back[pi[:, 1]] = np.arange(pi.shape[0], dtype=np.uint32)
    return back


def frame(simulator):
    """(position[N,4] in orig order, density[N] in orig order) of the simulator's current state."""
    pos = simulator.ocl_solver.read_position_buffer()
    rho_sorted = simulator.getDensity_cpp()
    back = invert_particle_index(simulator.getParticleIndex_cpp())
    return pos, rho_sorted[back]


def density_colour(rho, rho0):
    """RGB per particle exactly as display() picks it (owWorldSimulation.cpp:127-141): float arithmetic, later ramps win."""
    rho = np.clip(np.asarray(rho, np.float32), np.float32(0), np.float32(2) * np.float32(rho0))
    rho0 = np.float32(rho0)
    rgb = np.zeros((rho.shape[0], 3), np.float32)
    rgb[:, 2] = 1.0  # blue
    hundred = np.float32(100)
    ramps = ((1.00, lambda dc: (0 * dc, dc, 1 + 0 * dc)), (1.01, lambda dc: (0 * dc, 1 + 0 * dc, 1 - dc)),
             (1.02, lambda dc: (dc, 1 + 0 * dc, 0 * dc)), (1.03, lambda dc: (1 + 0 * dc, 1 - dc, 0 * dc)),
             (1.04, lambda dc: (1 + 0 * dc, 0 * dc, 0 * dc)))
    for factor, colour in ramps:
        dc = hundred * (rho - rho0 * np.float32(factor)) / rho0
        m = dc > 0
        r, g, b = colour(dc[m])
        rgb[m, 0], rgb[m, 1], rgb[m, 2] = r, g, b
    return rgb  # like glColor4f, components outside [0,1] are left to the consumer to clamp


def write_npz(path, position, density, step=None):
    np.savez_compressed(path, position=np.asarray(position, np.float32), density=np.asarray(density, np.float32),
                        step=np.int64(-1 if step is None else step))


def write_vtk(path, position, density, include_boundary=False):
    """Legacy-VTK polydata (binary, big-endian): points + `density` and `type` point scalars."""
    pos = np.asarray(position, np.float32).reshape(-1, 4)
    rho = np.asarray(density, np.float32)
    if not include_boundary:
        keep = pos[:, 3].astype(np.int32) != 3  # BOUNDARY_PARTICLE (owOpenCLConstant.h:12)
        pos, rho = pos[keep], rho[keep]
    n = pos.shape[0]
    with open(path, "wb") as f:
        f.write(b"# vtk DataFile Version 3.0\nsphmi frame\nBINARY\nDATASET POLYDATA\n")
        f.write(("POINTS %d float\n" % n).encode())
        f.write(pos[:, :3].astype(">f4").tobytes())
        f.write(("\nVERTICES %d %d\n" % (n, 2 * n)).encode())
        cells = np.empty((n, 2), ">i4")
        cells[:, 0] = 1
        cells[:, 1] = np.arange(n)
        f.write(cells.tobytes())
        f.write(("\nPOINT_DATA %d\nSCALARS density float 1\nLOOKUP_TABLE default\n" % n).encode())
        f.write(rho.astype(">f4").tobytes())
        f.write(b"\nSCALARS type float 1\nLOOKUP_TABLE default\n")
        f.write(pos[:, 3].astype(">f4").tobytes())
        f.write(b"\n")
    return n
