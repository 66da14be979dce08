import torch
def frag_to_dense(buf, F, M):
    nblk = buf.numel() // (F * 128)
    x = buf.view(torch.bfloat16).reshape(nblk, F // 16, 2, 64, 8).float().cpu()
    out = torch.zeros(nblk * 64, F)
    for h in range(2):
        for j in range(8):
            f_in = 8 * (j >> 2) + 4 * h + (j & 3)
            # lanes h*32 .. h*32+31 -> samples ct*32 + c
            v = x[:, :, :, h * 32:(h + 1) * 32, j]            # [nblk, ks, ct, 32]
            for ks in range(F // 16):
                out[:, ks * 16 + f_in] = v[:, ks].reshape(nblk * 64)
    return out[:M]

