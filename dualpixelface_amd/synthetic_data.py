"""Synthetic dual-pixel samples with the batch contract of the reference loader (dataloader/FaceDP/loader.py:149-155):
left / right [3,H,W], disp / depth / idepth / mask [H,W], normal [3,H,W], K [3,3], abvalue [2] -- the SURVEY section 8d inputs,
one sample per index.  For smoke runs of the trainer and the bench when no FaceDP data is on disk."""
import torch.utils.data as torch_data

from .recipe import synthetic_batch


class SyntheticDP(torch_data.Dataset):
    def __init__(self, length=16, height=256, width=384, seed=0, mask_mode='ones'):
        self.length, self.height, self.width, self.seed, self.mask_mode = int(length), int(height), int(width), int(seed), mask_mode

    def __len__(self):
        return self.length

    def __getitem__(self, idx):
        batch = synthetic_batch(1, self.height, self.width, seed=self.seed * 100003 + int(idx), mask_mode=self.mask_mode)
        return {k: v[0] for k, v in batch.items()}


def synthetic_loader(length, height, width, batch_size, shuffle=False, seed=0, workers=0):
    return torch_data.DataLoader(SyntheticDP(length, height, width, seed), batch_size=batch_size, shuffle=shuffle, num_workers=workers,
                                 drop_last=False)
