import sys, cProfile, pstats, numpy as np, torch
sys.path.insert(0, '/root/repo')
import quake_amd as quake
g = torch.Generator().manual_seed(1)
n, d, nlist = 2_000_000, 64, 2000
cent = torch.randn(nlist, d, generator=g) * 4
x = cent[torch.randint(0, nlist, (n,), generator=g)] + torch.randn(n, d, generator=g)
idx = quake.QuakeIndex()
bp = quake.IndexBuildParams(); bp.nlist = nlist; bp.niter = 2
idx.build(x, torch.arange(n), bp)
mp = quake.MaintenancePolicyParams(); mp.window_size = 2048; mp.refinement_radius = 8; mp.refinement_iterations = 2
idx.initialize_maintenance_policy(mp); idx.track_hits = True
sp = quake.SearchParams(); sp.k, sp.nprobe = 10, 8
for i in range(3):
    idx.search(x[i * 1024:(i + 1) * 1024], sp)
idx.maintenance()
idx.search(x[:1024], sp)
pr = cProfile.Profile(); pr.enable()
for i in range(5):
    idx.search(x[i * 1024:(i + 1) * 1024], sp)
    idx.maintenance()
pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(18)
