import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import stereovision_amd as S
c = S.Context(0)
names = ["xor+bcnt","xor+add","sub+add3","xor+min","xor+dppadd","xor+pkadd16","xor+mad24","xor+lshladd"]
for m in range(8):
    print(names[m], ["%.1f" % (c.probe_valu_rate(m, w, 4000)/1e12) for w in (1,2,4,8)])
