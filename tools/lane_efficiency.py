import sys; sys.path.insert(0,'.')
import raytracing_amd as R
from raytracing_amd.host_scenes import HostScene
import numpy as np
w,h=1920,1080
hs=HostScene.random_spheres(w,h,1); sc=hs.upload()
acc=R.DeviceBuffer(w*h*3*4)
for depth in (1,2,50):
    st=sc.render(acc.ptr,w,h,0,8,depth,R.MODE_FAST_FP32,seed=1,flags=R.FLAG_COUNT_TESTS)
    r=st['rays']
    print('depth',depth,'rays',r,'bounce lane-eff %.2f'%(r/(64*st['wave_bounces'])),'node lane-eff %.2f'%(st['box_tests']/2/(64*st['wave_node_steps'])),'leaf lane-eff %.2f'%(st['leaf_tests']/(64*max(st['wave_leaf_steps'],1))),
      'wave node steps/ray %.2f'%(st['wave_node_steps']*64/r),'wave leaf steps/ray %.2f'%(st['wave_leaf_steps']*64/r), 'nodes/ray %.2f'%(st['box_tests']/2/r),'leaf/ray %.2f'%(st['leaf_tests']/r))
