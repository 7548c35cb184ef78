"""Offline study (CPU, numpy float32): can the gradient map's pinned fp32 expression (get_gradient_compute.glsl:5-23 as pinned in DESIGN.md section 3:
taps b / 255, the three differences in the shader's association, sqrt of the sum of squares, x 0.25 m, R8_UNORM round-to-nearest-even) be replaced by
something cheaper that gives the same byte for all 2^32 tap tuples?  Two candidates on 20 M random tuples each of two populations:
  * "variant7": the same fp32 chain with 7 instead of 9 additions (shared q0 - q1, q0 + q1, q3 - q2, q3 + q2);
  * "integer": N = Gx^2 + Gy^2 + Gz^2 from the BYTES (exact, <= 780 300), result = round-half-even(sqrt(N) / 4).
Result (profiles/r6_gradient_shortcuts.txt): both differ from the pinned chain - the integer rule ONLY on exact ties N = (4k + 2)^2, where the true
value is k + 0.5 and the fp32 chain's rounding residue decides the byte (a third of the ties go the other way); ties are 0.16 % of uniform tuples and
2 % of the tuples of a noisy flat region (bytes 0..20: G = (2, 0, 0) is a tie at k = 0), i.e. most 64-voxel waves of real data hold one.
usage: python tools/gradient_shortcut_study.py"""
import numpy as np
rng=np.random.default_rng(1)
n=20_000_000
b=rng.integers(0,256,size=(4,n),dtype=np.uint8)
# also a noisy-flat population (bytes 0..20)
b2=rng.integers(0,21,size=(4,n),dtype=np.uint8)
f32=np.float32
def pinned(b):
    q=(b.astype(f32)/f32(255.0)).astype(f32)
    q0,q1,q2,q3=q
    tx=q0-q1; ty=(-q0)-q1
    sx=(tx-q2)+q3; sy=(ty+q2)+q3; sz=((-q0+q1)-q2)+q3
    S=(sx*sx+sy*sy)+sz*sz
    ln=np.sqrt(S).astype(f32)
    g=ln*f32(0.25)
    return np.rint(np.clip(g,0,1)*f32(255.0)).astype(np.uint8), S
def variant7(b):
    q=(b.astype(f32)/f32(255.0)).astype(f32)
    q0,q1,q2,q3=q
    tx=q0-q1; tp=q0+q1; u=q3-q2; w=q3+q2
    sx=tx+u; sy=w-tp; sz=u-tx
    S=(sx*sx+sy*sy)+sz*sz
    ln=np.sqrt(S).astype(f32)
    return np.rint(np.clip(ln*f32(0.25),0,1)*f32(255.0)).astype(np.uint8)
def integer(b):
    i=b.astype(np.int64)
    gx=i[0]-i[1]-i[2]+i[3]; gy=-i[0]-i[1]+i[2]+i[3]; gz=-i[0]+i[1]-i[2]+i[3]
    N=gx*gx+gy*gy+gz*gz
    v=np.sqrt(N.astype(np.float64))/4.0
    return np.minimum(np.rint(v),255).astype(np.uint8), N
for name,bb in (("uniform bytes",b),("noise 0..20",b2)):
    p,S=pinned(bb); v7=variant7(bb); ig,N=integer(bb)
    r=np.sqrt(N); tie=(N%4==0)&(np.sqrt(N//4)%1==0)&((np.sqrt(N//4)).astype(np.int64)%2==1)
    print(name, "variant7 mismatches %.3e"%((p!=v7).mean()), "integer-RNE mismatches %.3e"%((p!=ig).mean()), "ties %.3e"%tie.mean(), "mismatch&~tie %.3e"%(((p!=ig)&~tie).mean()))
