python - <<'EOP'
import sys, os
sys.path.insert(0, os.getcwd())
import torch, odtk
from odtk import ops
import subprocess
EOP
python tools/conv_bench.py conv1_2,conv2_2,conv3_2,conv4_2,conv5_2,conv7 fwd 10 2,2128,2,2128
