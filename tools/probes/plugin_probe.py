"""The drop-in path alone (bench.measure_plugin_path) at one configuration: python tools/probes/plugin_probe.py [c3|c2|c4] """
import json
import sys

sys.path.insert(0, ".")
import bench  # noqa: E402
import pydeseq2_amd  # noqa: E402
from pydeseq2_amd._lib import Context  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "c3"
G, N, design = bench.CONFIGS[cfg]
counts, X = bench.synth_fast(G, N, design, seed=bench.SEEDS[cfg])
ctx = Context(0)
res = pydeseq2_amd.DeseqPipeline(counts, X, ctx=ctx).deseq2()
out = bench.measure_plugin_path(counts, X, ctx, res)
print(json.dumps(out))
