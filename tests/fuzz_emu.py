"""Randomised stress of the persistent kernels on the CPU emulator (test infrastructure; not collected by pytest):

    python tests/fuzz_emu.py <seed> <iterations>

Random shapes for the halo conv (tiles 12 / 13 / 17 / 34), the LDS-DMA igemm (one tile per workgroup and persistent stream,
plain and gathered), the wide-tile conv (tiles 41 / 42: GroupNorm prologue, residual rows fetched by the last slab, one to three
slabs) and the wide GEMM (tiles 51 - 56: plain / two sources, the 3x3 im2col gather with stride 1 / 2 and padding 0 / 1, K slices,
GroupNorm partial sums) under randomly chosen schedules of tests/emu (I2I_EMU_ASYNC, I2I_EMU_ORDER) and workgroup counts
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
    kind=rnd.choice(["halo","gemm","gemm_gather","w32conv","g32","g32_gather","g32_stats"])
    dt=rnd.choice([torch.bfloat16, torch.float16, torch.float32])
    if kind in ("w32conv","g32","g32_gather","g32_stats") and dt==torch.float32: dt=torch.bfloat16      # 16-bit kernels
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
        elif kind=="gemm_gather":
            cin=rnd.choice([64,128]) if dt!=torch.float32 else rnd.choice([32,64])
            kw=dict(n=rnd.choice([1,2]), cin=cin, cout=rnd.choice([40,72,136]), h=rnd.choice([8,12]), w=rnd.choice([10,16]),
                    stride=rnd.choice([1,2]), pad=1, tile=rnd.choice([22,24,25]))
            oc.check_conv(lib,"cpu",dt,**kw)
        elif kind=="w32conv":
            kw=dict(n=rnd.choice([1,2]), cin=rnd.choice([64,128,192]), cout=rnd.choice([128,136,256]), h=rnd.choice([8,9,16,20,33]), w=rnd.choice([32,40,65]),
                    gn=rnd.choice([True,False]), res=rnd.choice([True,True,False]), alpha=rnd.choice([1.0,0.5]), tile=rnd.choice([41,42]), seed=it)
            if kw["gn"]: kw["act"]=1
            oc.check_conv(lib,"cpu",dt,**kw)
        elif kind=="g32":
            two=rnd.choice([False,True])
            kw=dict(n=rnd.choice([1,2]), cin=rnd.choice([64,128,320]), cout=rnd.choice([96,128,160,200,328]), h=rnd.choice([5,8,13]), w=rnd.choice([16,23,37]),
                    ks=1, pad=0, res=rnd.choice([True,False]), bias=rnd.choice([True,False]), tile=rnd.choice([51,52,53,54,55,56]), seed=it)
            if two: kw["cin2"]=rnd.choice([64,128])
            oc.check_conv(lib,"cpu",dt,**kw)
        elif kind=="g32_gather":
            asym=rnd.choice([False,True])
            kw=dict(n=rnd.choice([1,2,3]), cin=rnd.choice([64,128]), cout=rnd.choice([96,128,160,168]), h=rnd.choice([5,8,9,12]), w=rnd.choice([6,7,10,13]),
                    res=rnd.choice([True,False]), tile=rnd.choice([51,52,53,54]), seed=it)
            if asym: kw.update(stride=2, asym_pad=True)
            else: kw.update(stride=rnd.choice([1,2]), pad=1)
            if rnd.choice([True,False]): kw["splitk"]=rnd.choice([2,3,4,5])
            oc.check_conv(lib,"cpu",dt,**kw)
        else:
            cfg=rnd.choice([53,54]); bm=256 if cfg==53 else 128
            ks=rnd.choice([1,3]); stride=rnd.choice([1,2]) if ks==3 else 1
            h,w=rnd.choice([(16,16),(32,16),(32,32)])
            if (h//stride)*(w//stride)%bm: h,w=32,32
            if (h//stride)*(w//stride)%bm: stride=1
            kw=dict(n=rnd.choice([1,2]), cin=rnd.choice([64,128]), cout=rnd.choice([128,256]), h=h, w=w, groups=rnd.choice([8,32]), tile=cfg, ks=ks, stride=stride,
                    res=rnd.choice([True,False]), seed=it)
            oc.check_conv_gn_part(lib,"cpu",dt,**kw)
    except AssertionError as e:
        fails+=1
        print("FAIL", kind, dt, kw, {k:os.environ.get(k) for k in ("I2I_EMU_ASYNC","I2I_EMU_ORDER","I2I_PERSIST_WGS")}, str(e)[:80])
print("done, fails =", fails)
