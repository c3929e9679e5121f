#!/bin/bash
# fc_refactor time against FC_FE_HUGE_NF (levels whose largest front has at least this order take 128-column block steps)
for w in 100000 3072 1536 1024 600 256; do echo "FC_FE_HUGE_NF=$w"; FC_FE_HUGE_NF=$w timeout -k 10 200 python scripts/refactor_time.py ${MESHES:-O1 mesh_middle_gmsh cavity_fine} 2>&1 | grep "fc_refactor ms"; done
