# generate inline-asm variants of the 64-sample recurrence for a microbenchmark
# registers (by constraint index): %0 y1(lo/hi pair v) %1 y2 ; temps; we use explicit vreg names inside asm via clobbers
def chain(variant):
    L=[]
    # fixed regs: y pairs: v[10:11], v[12:13], v[14:15] rotating; temps r=v[16:17], q=v[18:19]
    # inputs Pk,B2k: prefetch ring of D entries at v[20+4*d ..] (P lo,hi,B2 lo,hi) -> ds_read_b128 gives (x=P, y=B2)
    D=8
    Y=lambda k: "v[%d:%d]"%(10+2*(k%3),11+2*(k%3))
    ins=lambda k: 20+4*(k%D)
    reads = variant in ("lds128","lds128_narrow","lds64x2","ldsread2")
    if variant.startswith("sload"):
        pass
    if reads:
        for k in range(D):
            L.append(rd(variant,k,ins(k)))
    for k in range(64):
        if reads:
            L.append("s_waitcnt lgkmcnt(%d)"%(min(D-1,63-k)*(2 if variant=="lds64x2" else 1) + (0)))
        if reads:
            P="v[%d:%d]"%(ins(k),ins(k)+1); B="v[%d:%d]"%(ins(k)+2,ins(k)+3)
        elif variant.startswith("sload"):
            blk=k//4; sb=40+16*(blk%2); P="s[%d:%d]"%(sb+4*(k%4),sb+4*(k%4)+1); B="s[%d:%d]"%(sb+4*(k%4)+2,sb+4*(k%4)+3)
            if k%4==0:
                L.append("s_waitcnt lgkmcnt(0)")
                nb=blk+1
                if nb<16:
                    L.append("s_load_dwordx16 s[%d:%d], %%[sbase], 0x%x"%(40+16*(nb%2),40+16*(nb%2)+15, 64*nb))
        else:
            P="v[20:21]"; B="v[22:23]"
        y1=Y(k-1); y2=Y(k-2); yn=Y(k)
        L.append("v_mul_f64 v[16:17], %%[a1], %s"%y1)
        L.append("v_add_f64 v[16:17], %s, v[16:17]"%B)
        L.append("v_mul_f64 v[18:19], %%[a2], %s"%y2)
        L.append("v_add_f64 v[16:17], v[16:17], %s"%P)
        L.append("v_add_f64 %s, v[16:17], v[18:19]"%yn)
        if reads and k+D<64:
            L.append(rd(variant,k+D,ins(k+D)))
        if variant in ("narrow","lds128_narrow","sload_narrow"):
            L.append("s_lshl_b64 exec, exec, 1")
        if variant in ("lds128w",):
            pass
    if variant in ("narrow","lds128_narrow","sload_narrow"):
        L.append("s_mov_b64 exec, -1")
    return "\n".join('"%s\\n"'%x for x in L)
def rd(variant,k,reg):
    if variant in ("lds128","lds128_narrow"):
        return "ds_read_b128 v[%d:%d], %%[lbase] offset:%d"%(reg,reg+3,16*k)
    if variant=="lds64x2":
        return "ds_read_b64 v[%d:%d], %%[lbase] offset:%d\\n\"\n\"ds_read_b64 v[%d:%d], %%[lbase] offset:%d"%(reg,reg+1,16*k,reg+2,reg+3,16*k+8)
    if variant=="ldsread2":
        return "ds_read2_b64 v[%d:%d], %%[lbase] offset0:%d offset1:%d"%(reg,reg+3,2*k,2*k+1)
import sys
out=open('/tmp/rb/variants.h','w')
for v in ("bare","narrow","lds128","lds128_narrow","lds64x2","ldsread2","sload","sload_narrow"):
    out.write("#define ASM_%s \\\n"%v.upper())
    out.write(" \\\n".join(chain(v).split("\n")))
    out.write("\n\n")
