import numpy as np
rng = np.random.default_rng(1)
def ray_steps(n):
    return np.minimum(1 + rng.gamma(shape=3.0, scale=4.7, size=n).astype(int), 48)
def sim(n_rays, n_waves, bw, D, box_cap=64, adopt_min_idle=8):
    pool = n_rays
    rem = np.zeros((n_waves, 64), int)
    exited = np.zeros(n_waves, bool)
    box = [[] for _ in range(n_waves // max(bw, 1))]
    total = 0; dry = 0; t = 0; donations = 0
    while True:
        live = (rem > 0).sum(axis=1)
        if pool > 0:
            for w in np.where((live <= 44) & ~exited)[0]:
                k = min(64 - live[w], pool)
                if k <= 0: continue
                idx = np.where(rem[w] == 0)[0][:k]
                rem[w, idx] = ray_steps(k); pool -= k
        elif bw > 1:
            for b in range(len(box)):
                ws = [w for w in range(b * bw, (b + 1) * bw) if not exited[w]]
                # adopt first (waves with many live lanes and idle room), then donate (sparse waves, if a peer stays alive)
                for w in sorted(ws, key=lambda w: -live[w]):
                    idle = 64 - (rem[w] > 0).sum()
                    if box[b] and idle >= adopt_min_idle and (rem[w] > 0).sum() > 0:
                        k = min(idle, len(box[b])); idx = np.where(rem[w] == 0)[0][:k]
                        rem[w, idx] = box[b][:k]; del box[b][:k]
                alive = [w for w in ws]
                for w in sorted(ws, key=lambda w: live[w]):
                    lv = rem[w][rem[w] > 0]
                    if 0 < len(lv) <= D and len(alive) > 1 and len(box[b]) + len(lv) <= box_cap:
                        box[b].extend(lv.tolist()); rem[w] = 0; exited[w] = True; alive.remove(w); donations += 1
                # a wave that ran empty takes the box over if nobody else can (the last wave never leaves rays behind)
                for w in ws:
                    if not exited[w] and (rem[w] > 0).sum() == 0:
                        if box[b]:
                            k = min(64, len(box[b])); rem[w, :k] = box[b][:k]; del box[b][:k]
                        else:
                            exited[w] = True
        live = (rem > 0).sum(axis=1)
        running = (live > 0) & ~exited
        if pool == 0 and not running.any() and not any(box): break
        n_run = int(running.sum()) if pool == 0 else int((~exited).sum())
        total += n_run
        if pool == 0: dry += n_run
        rem = np.maximum(rem - 1, 0)
        t += 1
        if t > 5000: break
    return total, dry, t, donations
base = sim(1_360_000, 6144, 1, 0)
print("1-wave blocks: wave-steps %d, after dry %d, makespan %d" % base[:3])
for bw in (4, 8):
    for D in (8, 16, 24, 32):
        tot, dry, t, don = sim(1_360_000, 6144, bw, D)
        print("%d-wave blocks, donate at <= %2d live: wave-steps %d (%.1f %%), after dry %d, makespan %d, donations %d" % (bw, D, tot, 100.0 * tot / base[0] - 100, dry, t, don))
