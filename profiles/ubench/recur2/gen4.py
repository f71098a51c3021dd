# DPP chain: inputs from row-replicated register sets via v_fmac_f64_dpp row_newbcast (exact add: x + in*1.0),
# outputs captured by v_mov_b64_dpp with row/bank masks.
# registers: Y pairs v[10:17] (y(k) in pair k&3), T v[18:19], Q v[20:21], ONE v[22:23],
#   input sets j=0..3: P_j v[24+4j:25+4j], B_j v[26+4j:27+4j]   (lane 16r+i holds sample 16j+i)
#   capture Z_m (m=0..3) v[40+2m:41+2m]: lane n = 16r + 4m + b gets y(n) in Z_m
import sys
def gen(capture=True, cap_delay=2):
    L=[]
    Y=lambda k: "v[%d:%d]"%(10+2*(k&3),11+2*(k&3))
    pend=[]
    for k in range(64):
        j=k>>4; i=k&15
        P="v[%d:%d]"%(24+4*j,25+4*j); B="v[%d:%d]"%(26+4*j,27+4*j)
        y1=Y(k-1); y2=Y(k-2); yn=Y(k)
        seq=["v_mul_f64 v[20:21], %%[a2], %s"%y2,
             "v_mul_f64 v[18:19], %%[a1], %s"%y1]
        # capture of an earlier sample here: >= 2 instructions after its add
        if capture and pend:
            seq.append(pend.pop(0))
        seq+=["v_fmac_f64_dpp v[18:19], %s, v[22:23] row_newbcast:%d row_mask:0xf bank_mask:0xf"%(B,i),
             "v_fmac_f64_dpp v[18:19], %s, v[22:23] row_newbcast:%d row_mask:0xf bank_mask:0xf"%(P,i),
             "v_add_f64 %s, v[18:19], v[20:21]"%yn]
        r=k>>4; b=(k>>2)&3; m=k&3
        pend.append("v_mov_b64_dpp v[%d:%d], %s row_newbcast:0 row_mask:0x%x bank_mask:0x%x"%(40+2*m,41+2*m,yn,1<<r,1<<b))
        L+=seq
    if capture:
        L.append("s_nop 1")
        L+=pend
    return L
def emit(name, L, out):
    out.write("#define %s \\\n"%name)
    out.write(" \\\n".join('"%s\\n"'%x for x in L))
    out.write("\n\n")
if __name__=="__main__":
    out=open(sys.argv[1],'w')
    emit("ASM_DPP", gen(True), out)
    emit("ASM_DPP_NOCAP", gen(False), out)
