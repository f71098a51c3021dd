# Generator of the hand-scheduled 64-sample recurrence (inline asm) -- variants for the microbenchmark and the product.
# Register plan (fixed names, all in the clobber list):
#   v[10:17]  Y0..Y3: y(k) lives in pair k&3      v[18:19] T   v[20:21] Q    v[24:87] input ring, 16 entries x {P lo,hi,B2 lo,hi}
import sys
def gen(D=12, narrow=True, group=4, sched=0, waits="group"):
    L=[]
    Y=lambda k: "v[%d:%d]"%(10+2*(k&3),11+2*(k&3))
    ring=lambda k: 24+4*(k%16)
    L.append("s_waitcnt lgkmcnt(0)")
    for k in range(D):
        L.append("ds_read_b128 v[%d:%d], %%[lbase] offset:%d"%(ring(k),ring(k)+3,16*k))
    for k in range(64):
        P="v[%d:%d]"%(ring(k),ring(k)+1); B="v[%d:%d]"%(ring(k)+2,ring(k)+3)
        if waits=="group":
            if k%group==0:
                last=min(63,k-1+D)   # last read issued so far
                need=min(63,k+group-1)
                L.append("s_waitcnt lgkmcnt(%d)"%max(0,last-need))
        else:
            last=min(63,k-1+D); L.append("s_waitcnt lgkmcnt(%d)"%max(0,last-k))
        rd = "ds_read_b128 v[%d:%d], %%[lbase] offset:%d"%(ring(k+D),ring(k+D)+3,16*(k+D)) if k+D<64 else None
        y1=Y(k-1); y2=Y(k-2); yn=Y(k)
        if sched==0:
            seq=["v_mul_f64 v[20:21], %%[a2], %s"%y2, "v_mul_f64 v[18:19], %%[a1], %s"%y1, rd, "v_add_f64 v[18:19], %s, v[18:19]"%B,
                 "v_add_f64 v[18:19], v[18:19], %s"%P, "v_add_f64 %s, v[18:19], v[20:21]"%yn]
        elif sched==1:
            seq=["v_mul_f64 v[18:19], %%[a1], %s"%y1, "v_mul_f64 v[20:21], %%[a2], %s"%y2, "v_add_f64 v[18:19], %s, v[18:19]"%B, rd,
                 "v_add_f64 v[18:19], v[18:19], %s"%P, "v_add_f64 %s, v[18:19], v[20:21]"%yn]
        else:
            seq=["v_mul_f64 v[18:19], %%[a1], %s"%y1, "v_add_f64 v[18:19], %s, v[18:19]"%B, "v_mul_f64 v[20:21], %%[a2], %s"%y2, 
                 "v_add_f64 v[18:19], v[18:19], %s"%P, rd, "v_add_f64 %s, v[18:19], v[20:21]"%yn]
        L += [x for x in seq if x]
        if narrow and k%group==group-1:
            L.append("s_lshl_b64 exec, exec, %d"%group)
    if narrow:
        L.append("s_mov_b64 exec, -1")
    return L
def emit(name, L, out):
    out.write("#define %s \\\n"%name)
    out.write(" \\\n".join('"%s\\n"'%x for x in L))
    out.write("\n\n")
if __name__=="__main__":
    out=open(sys.argv[1],'w')
    emit("ASM_N12_S0", gen(12,True,4,0), out)
    emit("ASM_N12_S1", gen(12,True,4,1), out)
    emit("ASM_N12_S2", gen(12,True,4,2), out)
    emit("ASM_N15_S0", gen(15,True,4,0), out)
    emit("ASM_W12_S0", gen(12,False,4,0), out)   # no narrowing (outputs ignored)
    emit("ASM_N12_S0_G8", gen(12,True,8,0), out)
    emit("ASM_N8_S0_PER", gen(8,True,4,0,"per"), out)
