"""Wall time of the text towers' recurrence at the client shape (B = 128 captions, T words, 300 -> 2 x 128): ops.bigru_last_states
(gru.hip) against the reference's lines on the library's packed GRU, forward + backward, same process."""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from creamfl_amd import ops, runtime  # noqa: E402
from torch.nn.utils.rnn import pack_padded_sequence, pad_packed_sequence


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=128)
    ap.add_argument('--words', type=int, default=30)
    ap.add_argument('--hidden', type=int, default=128)
    ap.add_argument('--iters', type=int, default=50)
    a = ap.parse_args()
    runtime.configure()
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(0)
    rnn = torch.nn.GRU(300, a.hidden, bidirectional=True, batch_first=True).to(dev)
    words = torch.randn(a.batch, a.words, 300, generator=g).to(dev).requires_grad_(True)
    lengths = torch.tensor(sorted(torch.randint(5, a.words + 1, (a.batch,), generator=g).tolist(), reverse=True))
    lengths_d = lengths.to(dev)
    gy = torch.randn(a.batch, 2 * a.hidden, generator=g).to(dev)

    def fused():
        (ops.bigru_last_states(rnn, words, lengths_d) * gy).sum().backward()

    def library():
        states, _ = pad_packed_sequence(rnn(pack_padded_sequence(words, lengths, batch_first=True))[0], batch_first=True)
        last = (lengths_d - 1).view(-1, 1, 1).expand(-1, 1, states.shape[2])
        (states.gather(1, last).squeeze(1) * gy).sum().backward()

    out = {'batch': a.batch, 'words': a.words, 'hidden': a.hidden, 'mean_len': float(lengths.float().mean())}
    for name, fn in (('fused', fused), ('library', library)):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.iters):
            fn()
        torch.cuda.synchronize()
        out[name + '_us'] = round((time.perf_counter() - t0) / a.iters * 1e6, 1)
    print(json.dumps(out))


if __name__ == '__main__':
    main()
