"""Regenerate the committed HIP model headers under sysbio_modeling_amd/csrc/models/."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sysbio_modeling_amd import build
from sysbio_modeling_amd.symbolic import zoo_model, ZOO_NAMES

for name in ZOO_NAMES:
    path = os.path.join(build.MODELS_DIR, name + '.hpp')
    with open(path, 'w') as fh:
        fh.write(zoo_model(name).hip_source)
    print("wrote", path)
