import os, sys, ctypes
sys.path.insert(0, os.getcwd())
order = sys.argv[1]
def maps():
    s=set()
    for line in open('/proc/self/maps'):
        if 'amdhip' in line or 'hsa-runtime' in line:
            s.add(line.split()[-1])
    return s
if order == 'torch_first':
    import torch
    print('torch avail', torch.cuda.is_available(), torch.cuda.device_count())
    x = torch.zeros(4, device='cuda'); print(x.sum().item())
    print(maps())
from nlos_surface_optimization_amd import _lib
l = _lib.lib()
print('maps after lib', maps())
print('count', l.nlos_device_count())
hip = ctypes.CDLL('libamdhip64.so.7')
n = ctypes.c_int(0)
rc = hip.hipGetDeviceCount(ctypes.byref(n)); print('direct hipGetDeviceCount rc', rc, n.value)
hip.hipGetErrorString.restype = ctypes.c_char_p
print(hip.hipGetErrorString(rc))
if order != 'torch_first':
    import torch
    print('torch avail', torch.cuda.is_available())
    x = torch.zeros(4, device='cuda'); print(x.sum().item())
    print(maps())
    print('count again', l.nlos_device_count())
