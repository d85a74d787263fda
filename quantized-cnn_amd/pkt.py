"""Probabilistic Knowledge Transfer (Passalis & Tefas, ECCV 2018; cited by the reference, README.md:24-29) as a loss
term for the quantized student: the divergence between the conditional probabilities that the batch's cosine-kernel
matrices of teacher and student features induce.  The reference's training loops hold no such term
(adp_qtz_hint.py:301-318 has no teacher pass); restated from the published method - parity unpinned.

    loss = criterion(logits, y) + model_loss(...) + pkt_weight * pkt_loss(student_features, teacher_features)
"""
import torch

from . import ops


class _PKT(torch.autograd.Function):
    @staticmethod
    def forward(ctx, student, teacher, eps):
        loss, d = ops.pkt_loss(student.contiguous(), teacher.detach().contiguous(), eps, need_grad=student.requires_grad)
        ctx.save_for_backward(d)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        (d,) = ctx.saved_tensors
        return (d * g if d is not None else None), None, None


def pkt_loss(student, teacher, eps=1e-7):
    """Scalar PKT divergence; differentiable with respect to `student` (N, Ds); `teacher` (N, Dt) is a constant."""
    return _PKT.apply(student, teacher, eps)
