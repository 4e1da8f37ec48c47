import sys, os
sys.path.insert(0,'.')
import numpy as np
import ngsdist_amd as N
def t(label):
    e=N.Engine(1000,1000000,kernel='mfma'); e.synth_fill(3)
    e.run(); a=[]
    for _ in range(4):
        e.run(); a.append(e.timing()['ms_accum'])
    print(label, 'accum ms', np.round(a,2)); del e
os.environ['NGD_DEBUG_POPULAR']='1'; t('plain layout, corner jobs on popular operands')
os.environ['NGD_DEBUG_POPULAR']='0'; t('plain layout')
