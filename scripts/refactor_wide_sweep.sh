#!/bin/bash
# fc_refactor time against the block-step threshold FC_FE_WIDE_NF (levels whose largest front has at least this order use 64-column steps)
for w in 3072 1500 1200 800 400; do echo "FC_FE_WIDE_NF=$w"; FC_FE_WIDE_NF=$w timeout -k 10 200 python scripts/refactor_time.py O1 mesh_middle_gmsh cavity_fine 2>&1 | grep "fc_refactor ms"; done
