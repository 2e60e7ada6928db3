"""Training steps through the VoxelFormer detector (host feature store -> pinned staging -> PCIe -> the lifting path ->
the reference's loss dict -> AdamW), bf16, B viewpoints per call: ms per step, against bench.py's device-resident step."""
import importlib, os, sys, tempfile, time
import numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests/golden')
pkg = importlib.import_module('vln-ver_amd'); syn = importlib.import_module('vln-ver_amd.synthetic')
config = importlib.import_module('vln-ver_amd.config')
importlib.import_module('vln-ver_amd.tuning').enable_tuned_gemms()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
dev = 'cuda'
torch.manual_seed(2)
det = pkg.build_detector(dict(config.load_model_cfg(), autocast_dtype='bf16', occupancy_rows=True))
det.pts_bbox_head.init_weights()
for k, p in det.named_parameters():
    if 'layout_branches.' in k or 'query_layout_embedding.' in k:
        p.requires_grad_(False)
det.to(dev).train()
root = tempfile.mkdtemp(prefix='ver_store_')
feats = syn.vit_features(B, seed=100)
w2p, org = syn.camera_batch(B, seed=1)
metas = []
for b in range(B):
    name = 'scanA_vp%d' % b
    for deg in range(6):
        np.save(os.path.join(root, '%s_i1_%d.npy' % (name, deg)), np.concatenate([np.zeros((1, 1, 768), np.float32), feats[b, deg][None]], 1))
    boxes, labels = syn.detection_gt(seed=40 + b, num_gt=3 + b % 5)
    rng = np.random.default_rng(7 + b)
    dense = rng.integers(0, 17, size=504000)
    dense[rng.uniform(size=504000) < 0.9] = 16               # ~10 % of the voxels occupied
    idx = np.nonzero(dense < 16)[0]
    occ_path = os.path.join(root, 'occ_%d.npy' % b)
    np.save(occ_path, np.stack([idx, dense[idx]], 1))
    metas.append(dict(sample_idx=name, file_name=root, occ_gt_path=occ_path, world2pixel=w2p[b], origin=org[b],
                      ann_info=dict(gt_bboxes_3d=torch.from_numpy(boxes[:, :7]), gt_labels_3d=labels)))
# one probing step: freeze what gets no gradient (the reference pays find_unused_parameters for it)
sum(det(return_loss=True, img_metas=metas[:2]).values()).backward()
for p in det.parameters():
    if p.requires_grad and p.grad is None:
        p.requires_grad_(False)
    p.grad = None
params = [p for p in det.parameters() if p.requires_grad]
opt = torch.optim.AdamW(params, lr=1e-4, weight_decay=0.01, fused=True)
def step():
    losses = det(return_loss=True, img_metas=metas)
    loss = sum(losses.values())
    loss.backward()
    torch.nn.utils.clip_grad_norm_(params, 300.0)
    opt.step(); opt.zero_grad(set_to_none=True)
    return loss
for _ in range(3): last = step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(steps): last = step()
torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / steps * 1e3
print('detector step: B=%d  %.1f ms per step  %.1f viewpoints/s  loss %.4f  peak %.1f GiB' % (B, ms, B / ms * 1e3, float(last.detach()), torch.cuda.max_memory_allocated() / 2**30))
