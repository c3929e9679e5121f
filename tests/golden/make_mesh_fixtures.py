"""Convert the mesh *data files* shipped with the reference into compact .npz fixtures.

Run in the build container (``/root/reference`` is not present on the GPU box):
    python tests/golden/make_mesh_fixtures.py
Inputs (data, not code): src/examples/{cylinder,cavity,pinball,lidcavity}/data_input/*.xdmf + .h5
Outputs: flowcontrol_amd/examples/<case>/data_input/<name>.npz (``flowcontrol_amd.examples.data.mesh_file``) with ``coords`` (nv,2) f8 and
``cells`` (nc,3) i4 in the file's own numbering, read with flowcontrol_amd's HDF5/XDMF reader.
"""
from pathlib import Path
import sys

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from flowcontrol_amd.fem.mesh import read_xdmf_mesh  # noqa: E402

REF = Path("/root/reference/src/examples")
MESHES = {
    "O1": REF / "cylinder/data_input/O1.xdmf",
    "cavity_coarse": REF / "cavity/data_input/cavity_coarse.xdmf",
    "cavity_fine": REF / "cavity/data_input/cavity_fine.xdmf",
    "mesh_middle_gmsh": REF / "pinball/data_input/mesh_middle_gmsh.xdmf",
    "lidcavity_mesh64": REF / "lidcavity/data_input/mesh64.xdmf",
}

if __name__ == "__main__":
    from flowcontrol_amd.examples.data import mesh_file

    for name, path in MESHES.items():
        m = read_xdmf_mesh(path, reorder=False)
        out = mesh_file(name)
        out.parent.mkdir(parents=True, exist_ok=True)
        np.savez_compressed(out, coords=m.coords, cells=m.cells.astype(np.int32))
        print(name, m.num_vertices, m.num_cells, out.stat().st_size)
