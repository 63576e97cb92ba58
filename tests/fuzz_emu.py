"""Randomised stress of the persistent kernels on the CPU emulator (test infrastructure; not collected by pytest):

    python tests/fuzz_emu.py <seed> <iterations>

Random shapes for the halo conv (tiles 12 / 13 / 17 / 34) and the LDS-DMA igemm (one tile per workgroup and persistent stream,
plain and gathered) under randomly chosen schedules of tests/emu (I2I_EMU_ASYNC, I2I_EMU_ORDER) and workgroup counts
(I2I_PERSIST_WGS).  Prints every failing configuration; 160 configurations (seeds 1-4 x 40) passed when this was written."""
import os, sys, random
HERE = os.path.dirname(os.path.abspath(__file__))
for _p in (os.path.dirname(HERE), HERE, os.path.join(HERE, 'emu')):
    sys.path.insert(0, _p)
import torch, build_emu, opcheck as oc
from img2img_turbo_amd import _capi
lib=_capi.Library(build_emu.build())
rnd=random.Random(int(sys.argv[1]) if len(sys.argv)>1 else 1)
fails=0
for it in range(int(sys.argv[2]) if len(sys.argv)>2 else 24):
    os.environ["I2I_EMU_ASYNC"]=str(rnd.choice([0,1]))
    o=rnd.choice([None,0,1,3,11])
    if o is None: os.environ.pop("I2I_EMU_ORDER",None)
    else: os.environ["I2I_EMU_ORDER"]=str(o)
    os.environ["I2I_PERSIST_WGS"]=str(rnd.choice([1,2,3,5,7]))
    kind=rnd.choice(["halo","gemm","gemm_gather"])
    dt=rnd.choice([torch.bfloat16, torch.float16, torch.float32])
    try:
        if kind=="halo":
            cin=rnd.choice([64,128,192]) if dt!=torch.float32 else rnd.choice([32,64,96])
            kw=dict(n=rnd.choice([1,2,3]), cin=cin, cout=rnd.choice([64,72,128,200,256,328]), h=rnd.choice([8,9,16,20,24]), w=rnd.choice([16,17,32,40]),
                    gn=rnd.choice([True,False]), res=rnd.choice([True,False]), tile=rnd.choice([12,13,17,34]))
            if kw["gn"]: kw["act"]=1
            oc.check_conv(lib,"cpu",dt,**kw)
        elif kind=="gemm":
            kw=dict(n=rnd.choice([1,2]), cin=rnd.choice([64,88,128,320]), cout=rnd.choice([64,72,136,200,256]), h=rnd.choice([8,12,19]), w=rnd.choice([16,23]),
                    ks=1, pad=0, res=rnd.choice([True,False]), tile=rnd.choice([22,23,24,25]))
            oc.check_conv(lib,"cpu",dt,**kw)
        else:
            cin=rnd.choice([64,128]) if dt!=torch.float32 else rnd.choice([32,64])
            kw=dict(n=rnd.choice([1,2]), cin=cin, cout=rnd.choice([40,72,136]), h=rnd.choice([8,12]), w=rnd.choice([10,16]),
                    stride=rnd.choice([1,2]), pad=1, tile=rnd.choice([22,24,25]))
            oc.check_conv(lib,"cpu",dt,**kw)
    except AssertionError as e:
        fails+=1
        print("FAIL", kind, dt, kw, {k:os.environ.get(k) for k in ("I2I_EMU_ASYNC","I2I_EMU_ORDER","I2I_PERSIST_WGS")}, str(e)[:80])
print("done, fails =", fails)
