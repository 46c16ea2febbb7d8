import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from tests.test_gpu_codec import small_cfg, engine, oracle_exact, rms, CR
dev = torch.device("cuda")
cfg = small_cfg(); W = CR.random_codec_weights(cfg, seed=3)
g = torch.Generator().manual_seed(1)
codes = torch.randint(0, cfg.codebook_size, (3, cfg.num_quantizers, 20), generator=g)
ref = oracle_exact(cfg, W, codes, 4)
dec = engine(cfg, W, dev, 3, 4); cache = dec.init_cache(3)
got = torch.cat([dec.decode_chunk(codes[:, :, t:t + 4], cache)[0].cpu().clone() for t in range(0, 20, 4)], -1).numpy()
print("small: rms err", rms(got - ref), "rms ref", rms(ref))
dec.close()
cfg = CR.CodecCfg(); W = CR.random_codec_weights(cfg, seed=0)
codes = torch.randint(0, 2048, (1, 16, 10), generator=g)
ex = oracle_exact(cfg, W, codes, 10)
dec = engine(cfg, W, dev, 2, 10); cache = dec.init_cache(1)
got = dec.decode_chunk(codes, cache)[0].cpu().numpy()
print("full: rms err", rms(got - ex), "rms ref", rms(ex))
