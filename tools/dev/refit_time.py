import time, numpy as np, loupiote_amd as lp
from loupiote_amd import scenes
dev=lp.Device(0)
desc=scenes.synthetic_atrium(textures=False); scene=scenes.to_product(desc)
t=time.perf_counter(); sg=lp.SceneGPU.new_from_scene(scene,dev); t_up=time.perf_counter()-t
c=scene.counts(); inst=scene.instances
ent=scene.entries
sizes=[ent[i["blas_index"]]["index_count"]//3 for i in inst]
idx=int(np.argmax(sizes))
for rep in range(3):
    m=inst[idx]["model_to_world"].reshape(-1).copy(); m[12]+=0.1*(rep+1)
    scene.set_instance_transform(idx,m)
    t=time.perf_counter(); n=sg.update_instances(scene); t_re=time.perf_counter()-t
    print("instance",idx,"tris",sizes[idx],"refit ms",round(t_re*1e3,2),"upload+build ms",round(t_up*1e3,1), "build_ms", sg.stats().build_ms)
