"""Host-side text output at `frequency` cadence: the formats of src/io.jl.

Out of the GPU scope -- kept minimal so run_simulation leaves
the same files behind as the reference: extended-XYZ (src/io.jl:42-70), LAMMPS dump with
unwrapped coordinates (src/io.jl:96-170), reader (src/io.jl:176-205), the log-spaced snapshot schedule
(src/io.jl:1-36) and zstd post-compression (src/io.jl:207-223, through pyarrow's zstd codec when present).
AsyncWriter takes the formatting and writing of a downloaded frame off the stepping thread.
"""
import os
import queue
import re
import threading

import numpy as np


def _g6(v):
    return "%.6g" % v


def write_to_file(filepath, step, unitcell, n_particles, positions, diameters, dimension, mode="a"):
    """src/io.jl:42-70"""
    unitcell = np.asarray(unitcell, dtype=np.float64)
    # Julia's comprehension [.. for i in 1:d, j in 1:d] iterates i fastest
    flat = " ".join(repr(float(unitcell[i, j])) for j in range(dimension) for i in range(dimension))
    pos = np.asarray(positions)
    with open(filepath, mode) as io:
        io.write(f"{n_particles}\n")
        io.write(f'Lattice="{flat}" Properties=type:I:1:id:I:1:radius:R:1:pos:R:{dimension} Time={_g6(step)}\n')
        cols = [np.ones(n_particles), np.arange(1, n_particles + 1), np.asarray(diameters) / 2.0]
        cols += [pos[:, d] for d in range(dimension)]
        fmt = "%d %d %f" + " %f" * dimension
        np.savetxt(io, np.column_stack(cols), fmt=fmt)


def write_to_file_lammps(filepath, step, unitcell, n_particles, positions, images, diameters, dimension, mode="w"):
    """src/io.jl:96-170"""
    boxmat = np.zeros((3, 3))
    boxmat[:dimension, :dimension] = np.asarray(unitcell)[:dimension, :dimension]
    pos = np.asarray(positions)
    img = np.asarray(images)
    with open(filepath, mode) as io:
        io.write("ITEM: TIMESTEP\n%d\n" % step)
        io.write("ITEM: NUMBER OF ATOMS\n%d\n" % n_particles)
        if dimension == 2:
            lx, ly = np.linalg.norm(boxmat[:, 0]), np.linalg.norm(boxmat[:, 1])
            io.write("ITEM: BOX BOUNDS xy pp pp\n")
            io.write("%f %f %f\n" % (0.0, lx, boxmat[0, 1]))
            io.write("%f %f 0.0\n" % (0.0, ly))
            io.write("%f %f 0.0\n" % (0.0, 1.0))
            io.write("ITEM: ATOMS id type radius x y xu yu\n")
        elif dimension == 3:
            io.write("ITEM: BOX BOUNDS xy xz yz pp pp pp\n")
            io.write("%f %f %f\n" % (0.0, np.linalg.norm(boxmat[:, 0]), boxmat[0, 1]))
            io.write("%f %f %f\n" % (0.0, np.linalg.norm(boxmat[:, 1]), boxmat[1, 2]))
            io.write("%f %f %f\n" % (0.0, np.linalg.norm(boxmat[:, 2]), boxmat[0, 2]))
            io.write("ITEM: ATOMS id type radius x y z xu yu zu\n")
        else:
            raise ValueError(f"Unsupported dimension: {dimension}")
        p3 = np.zeros((n_particles, 3))
        p3[:, :dimension] = pos
        i3 = np.zeros((n_particles, 3))
        i3[:, :dimension] = img
        uw = p3 + i3 @ boxmat.T
        cols = [np.arange(1, n_particles + 1), np.ones(n_particles), np.asarray(diameters) / 2.0]
        cols += [pos[:, d] for d in range(dimension)] + [uw[:, d] for d in range(dimension)]
        np.savetxt(io, np.column_stack(cols), fmt="%d %d" + " %f" * (1 + 2 * dimension))


def read_file(filepath, dimension=3):
    """src/io.jl:176-205 -> (unitcell, positions (N,d), diameters)"""
    with open(filepath) as io:
        n = int(io.readline())
        header = io.readline()
        m = re.search(r'Lattice="([^"]+)"', header)
        if m is None:
            raise ValueError("Could not parse Lattice property in file header")
        entries = np.array([float(t) for t in m.group(1).split()])
        unitcell = entries.reshape(dimension, dimension).T.copy()  # Julia reshape is column-major
        data = np.loadtxt(io, max_rows=n, ndmin=2)
    radii = data[:, 2]
    positions = data[:, 3:3 + dimension].copy()
    return unitcell, positions, radii * 2.0


def open_files(pathname, traj_name, thermo_name):
    """src/io.jl:225-239"""
    trajectory_file = os.path.join(pathname, traj_name)
    thermo_file = os.path.join(pathname, thermo_name)
    for f in (trajectory_file, thermo_file):
        if os.path.isfile(f):
            os.remove(f)
    return trajectory_file, thermo_file


def save_log_times_to_file(logs, logn, logbase, filename):
    """src/io.jl:1-15"""
    with open(filename, "w") as f:
        f.write(f"#maxsnap={logn},base={logbase}\n")
        for v in logs:
            f.write(f"{int(v)}\n")


def generate_log_times(max_iter=10000, logn=40, logbase=1.35, filename="new-log-times.txt"):
    """src/io.jl:17-36: floor(j*floor(base^logn) + base^i) for j in 0..max_iter, i in 0..logn, unique and sorted;
    also written to `filename` (the reference writes "new-log-times.txt" into the working directory)."""
    maxlog = int(np.floor(logbase ** logn))
    j = np.arange(max_iter + 1, dtype=np.float64)[:, None]
    i = np.arange(logn + 1, dtype=np.float64)[None, :]
    logs = np.unique(np.floor(j * maxlog + logbase ** i).astype(np.int64))
    if filename:
        save_log_times_to_file(logs, logn, logbase, filename)
    return [int(v) for v in logs]


def compress_zstd(filepath):
    """src/io.jl:207-223: <file> -> <file>.zst (a standard zstd frame), the original removed."""
    try:
        import pyarrow as pa
    except ImportError as e:                                        # pragma: no cover
        raise NotImplementedError("zstd compression needs pyarrow (its zstd codec) in this build") from e
    if not pa.Codec.is_available("zstd"):                           # pragma: no cover
        raise NotImplementedError("this pyarrow has no zstd codec")
    out = filepath + ".zst"
    with open(filepath, "rb") as src, pa.CompressedOutputStream(out, "zstd") as dst:
        while True:
            chunk = src.read(1 << 24)
            if not chunk:
                break
            dst.write(chunk)
    os.remove(filepath)


class AsyncWriter:
    """One background thread that runs queued write jobs in order (at most `depth` frames in flight), so that
    formatting a frame of text -- seconds at a million particles -- overlaps the next segment of steps.
    Exceptions of a job are re-raised by the next submit() or by close()."""

    def __init__(self, depth=2):
        self._q = queue.Queue(maxsize=depth)
        self._err = None
        self._t = threading.Thread(target=self._loop, daemon=True)
        self._t.start()

    def _loop(self):
        while True:
            job = self._q.get()
            if job is None:
                return
            fn, args, kw = job
            try:
                if self._err is None:
                    fn(*args, **kw)
            except BaseException as e:                              # noqa: BLE001
                self._err = e

    def submit(self, fn, *args, **kw):
        if self._err is not None:
            raise self._err
        self._q.put((fn, args, kw))

    def close(self):
        self._q.put(None)
        self._t.join()
        if self._err is not None:
            raise self._err
