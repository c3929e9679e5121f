"""Field checkpoints as XDMF + HDF5 (counterpart of the reference's ``utils/io.py:21-50``
``write_xdmf`` / ``read_xdmf`` built on ``dolfin.XDMFFile.write_checkpoint``).

``<file>.xdmf`` is a standard XDMF-3 temporal collection that ParaView opens: the P1 triangle mesh and,
per frame, the field sampled at the mesh vertices.  The payload is written with the pure-Python minimal
HDF5 writer (``fem/hdf5_min.py``):

    <file>.h5            /Mesh/mesh/geometry (nv,2) f8, /Mesh/mesh/topology (nc,3) i8     (once per series)
    <file>.<counter>.h5  /<name>/vector         full DoF vector in flowcontrol_amd's numbering (restart payload)
                         /<name>/vertex_values  (nv, ncomp) values at the vertices (visualisation)
                         /<name>/time           (1,)

One small file per frame makes appending a checkpoint O(frame) instead of rewriting the series.
The HDF5 layout is *ours* (dolfin's checkpoint layout stores its own dof numbering and cannot be
produced without dolfin); restart files written by the reference are therefore not readable here and
vice versa — only the mesh part and the vertex fields are interoperable.
"""

from __future__ import annotations

from pathlib import Path

import numpy as np

from .fem.hdf5_min import read_hdf5_tree, write_hdf5
from .fem.spaces import Function


def _h5(path: Path) -> Path:
    """Mesh file of a series (written once)."""
    return Path(path).with_suffix(".h5")


def _h5_frame(path: Path, counter: int) -> Path:
    """One small HDF5 file per frame: appending a checkpoint costs the frame, not the series so far."""
    return Path(path).with_suffix(f".{counter}.h5")


def _frame_times(path: Path) -> list[float]:
    """Times of the frames already listed in the .xdmf index."""
    import re

    if not Path(path).exists():
        return []
    return [float(m) for m in re.findall(r'<Time Value="([^"]+)"/>', Path(path).read_text())]


def _vertex_values(func: Function) -> np.ndarray:
    th = func.function_space().th
    a = func.vector().array()
    k = func.function_space().kind
    if k == "P":
        return a[: th.nv].reshape(-1, 1).copy()
    comps = [a[: th.nv], a[th.nn : th.nn + th.nv]]
    if k == "W":
        comps.append(a[2 * th.nn :])
    return np.stack(comps, axis=1)


def _xml(path: Path, name: str, th, times: list[float], ncomp: int) -> str:
    mesh_h5 = _h5(path).name
    nv, nc = th.nv, th.nc
    grids = []
    for c, t in enumerate(times):
        h5 = _h5_frame(path, c).name
        if ncomp == 1:
            att = f'<Attribute Name="{name}" AttributeType="Scalar" Center="Node"><DataItem Dimensions="{nv} 1" Format="HDF">{h5}:/{name}/vertex_values</DataItem></Attribute>'
        else:
            # a 2-D vector as two node scalars: understood by every XDMF reader
            att = (f'<Attribute Name="{name}_x" AttributeType="Scalar" Center="Node"><DataItem ItemType="HyperSlab" Dimensions="{nv} 1"><DataItem Dimensions="3 2" Format="XML">0 0 1 1 {nv} 1</DataItem>'
                   f'<DataItem Dimensions="{nv} {ncomp}" Format="HDF">{h5}:/{name}/vertex_values</DataItem></DataItem></Attribute>'
                   f'<Attribute Name="{name}_y" AttributeType="Scalar" Center="Node"><DataItem ItemType="HyperSlab" Dimensions="{nv} 1"><DataItem Dimensions="3 2" Format="XML">0 1 1 1 {nv} 1</DataItem>'
                   f'<DataItem Dimensions="{nv} {ncomp}" Format="HDF">{h5}:/{name}/vertex_values</DataItem></DataItem></Attribute>')
        grids.append(
            f'<Grid Name="{name}_{c}" GridType="Uniform"><Time Value="{t:.16g}"/>'
            f'<Topology TopologyType="Triangle" NumberOfElements="{nc}"><DataItem DataType="Int" Precision="8" Dimensions="{nc} 3" Format="HDF">{mesh_h5}:/Mesh/mesh/topology</DataItem></Topology>'
            f'<Geometry GeometryType="XY"><DataItem DataType="Float" Precision="8" Dimensions="{nv} 2" Format="HDF">{mesh_h5}:/Mesh/mesh/geometry</DataItem></Geometry>{att}</Grid>'
        )
    return ('<?xml version="1.0"?><Xdmf Version="3.0"><Domain><Grid Name="TimeSeries" GridType="Collection" CollectionType="Temporal">'
            + "".join(grids) + "</Grid></Domain></Xdmf>")


def write_xdmf(filename, func: Function, name: str, time_step: float = 0.0, append: bool = False, write_mesh: bool = True) -> int:
    """Append (or start) a checkpoint series; returns the frame counter written.

    Layout: ``<stem>.xdmf`` (index, rewritten: a few hundred bytes per frame), ``<stem>.h5`` (mesh, written
    when the series starts) and ``<stem>.<counter>.h5`` (dof vector, vertex values and time of one frame)."""
    path = Path(filename)
    path.parent.mkdir(parents=True, exist_ok=True)
    th = func.function_space().th
    times = _frame_times(path) if append else []
    if not times or not _h5(path).exists():
        write_hdf5(_h5(path), {"Mesh": {"mesh": {"geometry": th.mesh.coords, "topology": th.mesh.cells.astype(np.int64)}}})
    counter = len(times)
    vv = _vertex_values(func)
    write_hdf5(_h5_frame(path, counter), {name: {"vector": func.vector().get_local(), "vertex_values": vv, "time": np.array([float(time_step)])}})
    times.append(float(time_step))
    path.write_text(_xml(path, name, th, times, vv.shape[1]))
    return counter


def read_xdmf(filename, func: Function, name: str, counter: int = -1) -> float:
    """Load frame ``counter`` (−1: last) of series ``name`` into ``func``; returns its time."""
    path = Path(filename)
    n = len(_frame_times(path))
    c = n - 1 if counter < 0 else counter
    if c < 0 or c >= n or not _h5_frame(path, c).exists():
        raise FileNotFoundError(f"{filename}: series has no frame {c} ({n} frames)")
    tree = read_hdf5_tree(_h5_frame(path, c))
    if name not in tree:
        raise KeyError(f"{filename}: no series {name!r}; have {sorted(tree)}")
    vec = tree[name]["vector"]
    if vec.size != func.vector().size():
        raise ValueError(f"{filename}: frame has {vec.size} dofs, function space has {func.vector().size()} (different mesh?)")
    func.vector().set_local(vec)
    return float(tree[name]["time"][0])


__all__ = ["write_xdmf", "read_xdmf"]
