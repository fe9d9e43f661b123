import threading, time, hashlib, os
d = os.urandom(1<<20)*4
def w():
    for _ in range(40): hashlib.sha256(d).digest()
for n in (1,8,16,32,64,128):
    t=time.time(); th=[threading.Thread(target=w) for _ in range(n)]; [x.start() for x in th]; [x.join() for x in th]; dt=time.time()-t
    print(n, round(dt,2), "s ->", round(n/dt,1), "units/s")
