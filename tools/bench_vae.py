import sys, time, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/acoustic-image-generation_amd")
from acimg.session import Session
from acimg.trainer_vae import TrainerVAE
from acimg.unet_vae import UNet, UNetSound
dev = torch.device("cuda:0")
for cls, B in ((UNet, 32), (UNetSound, 32)):
    sess = Session(dev)
    tr = TrainerVAE(cls(), learning_rate=1e-4, session=sess)
    g = tr._build_functions(batch_size=B)
    tr.model.initialize(seed=3)
    g.images.copy_(torch.rand(*g.images.shape))
    for _ in range(3): tr.train_step(sync=False)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): tr.train_step(sync=False)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    print(cls.__name__, "B=%d  %.2f ms/step  %.0f img/s  launches %d" % (B, dt * 1e3, B / dt, len(g.plan_train)))
