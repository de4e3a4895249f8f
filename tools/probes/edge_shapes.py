import sys; sys.path[:0]=['.','tests']
import numpy as np
from gym_d2d_amd import _native as nat
from gym_d2d_amd.simulator import Simulator
from oracle import d2d_oracle as orc
from sim_util import default_links, random_layout
rng=np.random.default_rng(1)
for (b,rbs,cues,dues) in ((3,100000,25,25),(2,5000,300,300),(1,1,0,1),(1,1,1,0),(5,3000,1024,1024),(2,70000,1000,1000),(4,1,1,1)):
    sim=Simulator(dict(num_rbs=rbs,num_cues=cues,num_due_pairs=dues,num_envs=b))
    pos=random_layout(rng,b,cues,dues); sim.set_positions(pos); sim.set_links(sim.default_link_keys())
    p=sim.config.num_pwr_actions
    raw=np.concatenate([rng.integers(0,rbs*p['cue'],(b,cues)),rng.integers(0,rbs*p['due'],(b,dues))],1).astype(np.int32)
    sim.handle.set_obs_mode(nat.OBS_LINEAR if cues+dues<=128 else nat.OBS_TABLE)
    sim.step_arrays(raw)
    tx,rx,ty=default_links(cues,dues)
    ref=orc.full_step(pos.astype(np.float64),tx,rx,ty,raw,orc.device_columns(*orc.device_configs(cues,dues)[1:]),orc.PathLossSpec(),with_obs=False,chunk=2)
    got=sim.fetch(nat.BUF_SINR_DB).astype(np.float64)
    err=np.abs(got-ref['sinr_db'])/np.maximum(np.abs(ref['sinr_db']),1.0)
    print((b,rbs,cues,dues),'sinr err',float(err.max()),'reward err',float(np.abs(sim.fetch(nat.BUF_REWARD)[:,0]-ref['reward']).max()),'flags',sim.handle.status_flags())
    sim.handle.close()
