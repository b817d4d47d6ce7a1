import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import test_rconv_gpu as T
from mmgt_amd import hip
from mmgt_amd.packing import pack_rconv
def run(nb,H,W,c0,c1,cout,res,temb_rows,b2,tag,cbs=(320,256,160),reps=6):
    x0,x1,w,b,temb,r,scale,shift=T._case(nb,H,W,c0,c1,cout,100+nb+c0+c1,res=res,temb_rows=temb_rows)
    ref=T._ref(x0,x1,w,b,temb,max(b2,1),r,scale,shift)
    wimg=pack_rconv(w)
    for cb in cbs:
        if cout % cb: continue
        hip.tune("rconv_cb", cb)
        for rep in range(reps):
            out=torch.full((nb,H,W,cout), float("nan"), device=x0.device, dtype=torch.bfloat16)
            hip.gn_silu_conv3x3_unet(x0,scale,shift,wimg,cout,b,temb,b2,r,x1=x1,out=out)
            torch.cuda.synchronize()
            d=(out.double()-ref).abs()
            tol=ref.abs()*2.0**-8+9*(c0+c1)*2.0**-22*4
            bad=~(d<=tol)
            if bad.any():
                pi=bad.float().mean(dim=(1,2,3)); nz=pi.nonzero().flatten().tolist()
                rows=[i for i,v in enumerate(bad.float().mean(dim=(0,2,3)).tolist()) if v>0]
                cbad=bad.float().mean(dim=(0,1,2)); ch=(cbad>0).nonzero().flatten().tolist()
                print(tag,"cb",cb,"rep",rep,"bad images",nz[:12],"rows",rows,"channels",len(ch),ch[:4],ch[-4:], "nan", bool(torch.isnan(out.float()).any()), "max", float(d[~torch.isnan(d)].max()))
            else:
                print(tag,"cb",cb,"rep",rep,"ok")
    hip.tune("rconv_cb", 0)
run(300,16,16,320,0,320,True,0,0,"res",cbs=(160,))
run(300,16,16,320,0,320,False,0,0,"nores",cbs=(160,))
run(300,16,16,320,0,320,True,0,0,"res",cbs=(320,))
run(150,16,16,320,0,640,True,0,0,"res640",cbs=(160,320))
