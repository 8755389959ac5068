import sys, time, torch
sys.path.insert(0, "."); sys.argv = ["bench.py"]
import importlib.util
spec = importlib.util.spec_from_file_location("bench", "bench.py"); b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
import argparse
from gmlm_amd import bert
from transformers import BertConfig, BertModel
dev = torch.device("cuda")
enc = BertModel(BertConfig(vocab_size=30522, hidden_size=768, num_hidden_layers=12, num_attention_heads=12, intermediate_size=3072)).to(dev)
for _ in range(3): bert.prepare_weights(enc, torch.bfloat16)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20): bert.prepare_weights(enc, torch.bfloat16)
t1 = time.perf_counter(); torch.cuda.synchronize()
print("prepare_weights host time per call: %.0f us" % ((t1 - t0) / 20 * 1e6))
x = torch.zeros(5201, dtype=torch.bool); x[::4] = True
t0 = time.perf_counter()
for _ in range(100):
    i = x.nonzero(as_tuple=True)[0]; p = i.pin_memory().to(dev, non_blocking=True)
t1 = time.perf_counter()
print("nonzero+pin+h2d per call: %.0f us" % ((t1 - t0) / 100 * 1e6))
