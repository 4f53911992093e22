import time, numpy as np, sys, os
sys.path.insert(0,'.')
from gomokuai_amd import lib as G
import torch
n=int(os.environ.get('N','4096')); P=800
moves,lens,_=G.synth_boards(n,0)
lens=np.minimum(lens,4).astype(np.int32)
planes=G.moves_to_planes(moves,lens)
last=np.array([moves[i,lens[i]-1] for i in range(n)],dtype=np.int16)
R=int(os.environ.get('ROLLOUTS','5'))
t=G.BatchedMCTS(n,playouts_capacity=P,c_rollouts=R)
print('gpb env',os.environ.get('GMK_MCTS_GAMES_PER_BLOCK'),t.launch_info())
for rep in range(2):
    t.set_roots(planes,last,0)
    torch.cuda.synchronize()
    t0=time.perf_counter(); t.run(P); torch.cuda.synchronize(); dt=time.perf_counter()-t0
    v,q,rv,nodes,st=t.root_stats()
    print("rep",rep,"time %.1f ms"%(dt*1e3),"playouts/s %.3g"%(n*P/dt),"nodes mean",nodes.mean(),"status",st.any(),"bytes/playout",t.alg_bytes()/(n*P))
