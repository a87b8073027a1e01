"""BLAKE2Xs extendable-output function (host plumbing, not on the GPU path): what
``blake2s.NewXOF(blake2s.OutputLengthUnknown, nil)`` gives sign/bdn (bdn.go:29-33).  hashlib's
blake2s refuses the depth = 0 parameter block BLAKE2X needs for its output nodes, hence this small
parameterised BLAKE2s.  Checked against the coefficient fixtures of sign/bdn/bdn_vartime_test.go:24-33."""
import struct
IV=[0x6A09E667,0xBB67AE85,0x3C6EF372,0xA54FF53A,0x510E527F,0x9B05688C,0x1F83D9AB,0x5BE0CD19]
SIGMA=[[0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15],[14,10,4,8,9,15,13,6,1,12,0,2,11,7,5,3],[11,8,12,0,5,2,15,13,10,14,3,6,7,1,9,4],[7,9,3,1,13,12,11,14,2,6,5,10,4,0,15,8],[9,0,5,7,2,4,10,15,14,1,11,12,6,8,3,13],[2,12,6,10,0,11,8,3,4,13,7,5,15,14,1,9],[12,5,1,15,14,13,4,10,0,7,6,3,9,2,8,11],[13,11,7,14,12,1,3,9,5,0,15,4,8,6,2,10],[6,15,14,9,11,3,0,8,12,2,13,7,1,4,10,5],[10,2,8,4,7,6,1,5,15,11,9,14,3,12,13,0]]
M32=0xFFFFFFFF
def rotr(x,n): return ((x>>n)|(x<<(32-n)))&M32
def compress(h,block,t,last):
    m=list(struct.unpack('<16I',block)); v=h[:]+IV[:]
    v[12]^=t&M32; v[13]^=(t>>32)&M32
    if last: v[14]^=M32
    def G(a,b,c,d,x,y):
        v[a]=(v[a]+v[b]+x)&M32; v[d]=rotr(v[d]^v[a],16); v[c]=(v[c]+v[d])&M32; v[b]=rotr(v[b]^v[c],12)
        v[a]=(v[a]+v[b]+y)&M32; v[d]=rotr(v[d]^v[a],8); v[c]=(v[c]+v[d])&M32; v[b]=rotr(v[b]^v[c],7)
    for r in range(10):
        s=SIGMA[r]
        G(0,4,8,12,m[s[0]],m[s[1]]);G(1,5,9,13,m[s[2]],m[s[3]]);G(2,6,10,14,m[s[4]],m[s[5]]);G(3,7,11,15,m[s[6]],m[s[7]])
        G(0,5,10,15,m[s[8]],m[s[9]]);G(1,6,11,12,m[s[10]],m[s[11]]);G(2,7,8,13,m[s[12]],m[s[13]]);G(3,4,9,14,m[s[14]],m[s[15]])
    return [h[i]^v[i]^v[i+8] for i in range(8)]
def blake2s_param(data,digest_len,fanout,depth,leaf,node_off,xof_len,node_depth,inner):
    p=struct.pack('<BBBBIIHBB',digest_len,0,fanout,depth,leaf,node_off,xof_len,node_depth,inner)+bytes(16)
    h=[IV[i]^struct.unpack('<8I',p)[i] for i in range(8)]
    t=0
    blocks=[data[i:i+64] for i in range(0,len(data),64)] or [b""]
    for i,b in enumerate(blocks):
        lastb=(i==len(blocks)-1)
        t+=len(b)
        h=compress(h,b.ljust(64,b"\0"),t,lastb)
    return struct.pack('<8I',*h)[:digest_len]
def blake2xs(data,outlen,xof_len=0xFFFF):
    h0=blake2s_param(data,32,1,1,0,0,xof_len,0,0)
    out=b"";i=0
    while len(out)<outlen:
        rem=outlen-len(out)
        dl=32 if (xof_len==0xFFFF or rem>=32) else rem
        out+=blake2s_param(h0,dl,0,0,32,i,xof_len,0,32); i+=1
    return out[:outlen]
