"""``get_trainer`` with the reference's keyword surface (builder/trainer/__init__.py:14-47)."""
from .trainer import missing_to_num, missing_trainer  # noqa: F401


def get_trainer(args, iteration, x, static, input_lengths, y, output_lengths, model, logger, device, scheduler,
                optimizer, criterion, x_txt=None, x_img=None, txt_lengths=None, seq_lengths=None, imgtxt_time=None,
                scaler=None, missing=None, flow_type=None, reports_tokens=None, reports_lengths=None,
                criterion_aux=None):
    return missing_trainer(args, iteration, x, static, input_lengths, y, model, logger, device, scheduler, optimizer,
                           criterion, scaler, flow_type, output_lengths, seq_lengths=seq_lengths, x_img=x_img,
                           x_txt=x_txt, txt_lengths=txt_lengths, imgtxt_time=imgtxt_time, missing=missing,
                           reports_tokens=reports_tokens, reports_lengths=reports_lengths, criterion_aux=criterion_aux)
