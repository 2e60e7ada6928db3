"""Slot timeline of k_occ_mlp_bwd_ws (s_memtime stamps; library built with -DVER_WS_TIMELINE into scratch/r06/lib_ws_timeline.so):
per slot of the 8-slot round, how long each team WORKS (previous barrier's release -> its own last LDS operation complete) and
how long the slot lasts (release -> release), over 8 probe workgroups x 16 rounds.  Without the timeline library (VER_LIB = an
ablation build or the product library) only the launch time is printed.
    VER_LIB=scratch/r06/lib_ws_timeline.so python scratch/r06/ws_timeline.py [N]"""
import ctypes, importlib, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
hip = importlib.import_module('vln-ver_amd.hipops')
if os.environ.get('VER_LIB'):
    hip.LIB_PATH = os.path.abspath(os.environ['VER_LIB'])
N = int(sys.argv[1]) if len(sys.argv) > 1 else 96768000
dev = 'cuda'
g = torch.Generator(device='cpu').manual_seed(0)
def P(*s): return (torch.randn(*s, generator=g) * 0.1).to(dev).requires_grad_(True)
w2, b2, w3, b3 = P(128, 128), P(128), P(16, 128), P(16)
g1, be1, g2, be2 = (1 + P(128)).detach().requires_grad_(True), P(128), (1 + P(128)).detach().requires_grad_(True), P(128)
w2c = (w2 - w2.mean(0, keepdim=True)); b2c = b2 - b2.mean()
x = torch.randn(N // 8, 128, device=dev).repeat(8, 1)
x = (x - x.mean(1, keepdim=True)).to(torch.bfloat16).requires_grad_(True)
gy = (torch.randn(N, 16, device=dev) * 0.1).to(torch.bfloat16)
def step():
    out = hip.occ_mlp(x, None, None, g1, be1, w2c, b2c, g2, be2, w3, b3, centered=True)
    out.backward(gy)
    x.grad = None
for _ in range(2): step()
torch.cuda.synchronize()
timer = hip.KernelTimer(); hip.KERNEL_TIMER = timer
for _ in range(3): step()
hip.KERNEL_TIMER = None
kt = timer.summary()
ms = {k: v['ms'] / v['count'] for k, v in kt.items() if 'occ_mlp' in k}
print('rows %d lib %s: %s' % (N, os.path.basename(hip.LIB_PATH), ', '.join('%s %.2f ms' % kv for kv in ms.items())))
L = hip.lib()
if not hasattr(L, 'ver_ws_timeline_read'):
    sys.exit(0)
NP, NW, NR, NS = 8, 8, 16, 8
slots = (ctypes.c_longlong * (NP * NW * NR * NS * 2))(); span = (ctypes.c_longlong * (NP * NW * 2))()
rc = L.ver_ws_timeline_read(slots, span)
assert rc == 0, rc
t = np.array(list(slots), dtype=np.int64).reshape(NP, NW, NR, NS, 2)
sp = np.array(list(span), dtype=np.int64).reshape(NP, NW, 2)
bwd_ms = [v for k, v in ms.items() if 'backward' in k or 'bwd' in k][0]
life = (sp[:, :, 1] - sp[:, :, 0]).astype(np.float64)
tick_per_ms = life.mean() / bwd_ms                        # (persistent workgroups: a wave lives for the whole launch)
print('wave lifetime %.0f ticks avg (min %.0f max %.0f) over a %.2f-ms launch -> %.1f ticks/us' % (life.mean(), life.min(), life.max(), bwd_ms, tick_per_ms / 1e3))
nblk = (N + 63) // 64
rounds = nblk / 256 / 2
print('rounds per workgroup %.0f -> %.0f ticks (%.2f us) per round of two 64-row blocks' % (rounds, life.mean() / rounds, life.mean() / rounds / tick_per_ms * 1e3))
# per slot: start = release of the previous slot (which = 1), work end (which = 0), release (which = 1)
flat = t.reshape(NP, NW, NR * NS, 2)
start = flat[:, :, :-1, 1][:, :, NS - 1:]                  # release of slot i-1 for slots of rounds 1..NR-1
work_end = flat[:, :, 1:, 0][:, :, NS - 1:]
release = flat[:, :, 1:, 1][:, :, NS - 1:]
nslot = start.shape[2]
work = (work_end - start).reshape(NP, NW, nslot // NS, NS).astype(np.float64)
dur = (release - start).reshape(NP, NW, nslot // NS, NS).astype(np.float64)
names_row = ['r0(A) LN1 fwd', 'r6(B) LN1 bwd', 'r2(A) LN2 fwd', 'r0(B) LN1 fwd', 'r4(A) LN2 bwd', 'r2(B) LN2 fwd', 'r6(A) LN1 bwd', 'r4(B) LN2 bwd']
names_feat = ['f5(B) sums,dW2,dh1', 'f1(A) a2', 'f7(B) LN1 sums', 'f3(A) dh2,dW3', 'f1(B) a2', 'f5(A) sums,dW2,dh1', 'f3(B) dh2,dW3', 'f7(A) LN1 sums']
us = 1e3 / tick_per_ms
print('slot | row team: step, work ticks (mean of waves / slowest wave) | feature team: step, work ticks | slot ticks | idle: row, feature')
tot = np.zeros(5)
for s in range(NS):
    rw = work[:, :4, :, s]; fw = work[:, 4:, :, s]; d = dur[:, :, :, s]
    row_mean, row_max = rw.mean(), rw.max(axis=1).mean()
    f_mean, f_max = fw.mean(), fw.max(axis=1).mean()
    dd = d.mean()
    tot += [row_mean, row_max, f_mean, f_max, dd]
    print('%d | %-14s %6.0f / %6.0f | %-19s %6.0f / %6.0f | %6.0f | %5.0f %5.0f' % (s, names_row[s], row_mean, row_max, names_feat[s], f_mean, f_max, dd, dd - row_mean, dd - f_mean))
print('round| row work %6.0f / %6.0f | feature work %6.0f / %6.0f | slots %6.0f ticks = %.2f us' % (tot[0], tot[1], tot[2], tot[3], tot[4], tot[4] * us))
print('share of a round: row team working %.2f, feature team working %.2f; slot = max(slowest row wave, slowest feature wave) would be %.0f ticks'
      % (tot[0] / tot[4], tot[2] / tot[4], sum(max(work[:, :4, :, s].max(axis=1).mean(), work[:, 4:, :, s].max(axis=1).mean()) for s in range(NS))))

if hasattr(L, 'ver_ws_timeline_fine_read'):
    fine = (ctypes.c_longlong * (NP * NW * NR * 8))()
    assert L.ver_ws_timeline_fine_read(fine) == 0
    f = np.array(list(fine), dtype=np.int64).reshape(NP, NW, NR, 8)[:, :4, 1:]          # row team, rounds 1..
    rel = flat.reshape(NP, NW, NR, NS, 2)[:, :4, 1:]
    for name, base, slot in (('r4(A) LN2 bwd', 0, 4), ('r6(B) LN1 bwd', 4, 1)):
        t = f[..., base:base + 4].astype(np.float64)
        start = rel[..., slot - 1, 1].astype(np.float64)            # release of the previous slot
        end = rel[..., slot, 0].astype(np.float64)
        print('%s: entry %+5.0f | reads + masks + sums %5.0f | DPP reductions %5.0f | output %5.0f | to the end of the step %5.0f'
              % (name, (t[..., 0] - start).mean(), (t[..., 1] - t[..., 0]).mean(), (t[..., 2] - t[..., 1]).mean(),
                 (t[..., 3] - t[..., 2]).mean(), (end - t[..., 3]).mean()))
