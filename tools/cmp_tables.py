import sys,re
def load(f):
    d={}
    for l in open(f):
        p=l.split()
        if len(p)>10 and p[0] in('fwd','dgrad','wgrad'):
            d[(p[0],p[1])]=(float(p[8]), ' '.join(p[13:]))
    return d
a,b=load(sys.argv[1]),load(sys.argv[2])
ta=tb=0
for k in a:
    if k in b:
        ta+=a[k][0]; tb+=b[k][0]
        flag = '  <<<' if b[k][0] > a[k][0]*1.05 else ('  +' if b[k][0] < a[k][0]*0.95 else '')
        print('%-6s %-7s %.3f %-16s -> %.3f %-16s%s' % (k[0],k[1],a[k][0],a[k][1],b[k][0],b[k][1],flag))
print('total', ta, tb)
