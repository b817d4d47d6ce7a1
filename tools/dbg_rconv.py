import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import test_rconv_gpu as T
from mmgt_amd import hip
from mmgt_amd.packing import pack_rconv
def run(nb,H,W,c0,c1,cout,res,temb_rows,b2,tag):
    x0,x1,w,b,temb,r,scale,shift=T._case(nb,H,W,c0,c1,cout,100+nb+c0+c1,res=res,temb_rows=temb_rows)
    out=hip.gn_silu_conv3x3_unet(x0,scale,shift,pack_rconv(w),cout,b,temb,b2,r,x1=x1)
    torch.cuda.synchronize()
    ref=T._ref(x0,x1,w,b,temb,max(b2,1),r,scale,shift)
    d=(out.double()-ref).abs()
    tol=ref.abs()*2.0**-8+9*(c0+c1)*2.0**-22*4
    bad=(d>tol)
    print(tag,"bad frac",bad.float().mean().item(),"max",d.max().item())
    if bad.any():
        print(" per image:",bad.float().mean(dim=(1,2,3)).tolist())
        print(" per row y:",[round(v,3) for v in bad.float().mean(dim=(0,2,3)).tolist()])
        print(" per col x:",[round(v,3) for v in bad.float().mean(dim=(0,1,3)).tolist()])
        cb=bad.float().mean(dim=(0,1,2)); print(" per channel (first 40):",[round(v,2) for v in cb[:40].tolist()], "nonzero channels", int((cb>0).sum()))
run(1,16,16,320,0,320,False,0,0,"1 tile")
run(1,32,32,320,0,320,False,0,0,"4 tiles")
run(1,16,32,320,0,320,False,0,0,"1x2 tiles")
run(2,16,16,320,0,320,False,0,0,"2 images")
run(1,16,16,320,0,320,True,0,0,"res")
run(2,16,16,320,0,320,False,2,1,"temb")
run(3,32,48,320,0,320,True,3,1,"case2")
