"""cs/eval_finetune.py: evaluate the finetuned student alone (only 'model_student/*' is built and
restored, cs/eval_finetune.py:152-175,308-330; the logged loss is the student's cross entropy).

    python -m efficientvideoclassification_youtube8m_amd.eval_finetune --eval_data_pattern ... --train_dir \
        ./model_HLSTM_TeaStud_every10_finetune/ ... --run_once True                 # = run_eval.sh
"""
from . import validate


def main(argv=None):
    return validate.main(argv, student_only=True)


if __name__ == "__main__":
    main()
