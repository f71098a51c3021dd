cd /tmp; export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -oE "\b(TA_[A-Z_0-9a-z]+|TCP_[A-Z_0-9a-z]+|SQ_INSTS_[A-Z_0-9]+|SQ_INST_[A-Z_0-9]+|TD_[A-Z_0-9a-z]+|SQC_[A-Z_0-9a-z]+|SQ_IFETCH[A-Z_0-9]*|SQ_WAIT[A-Z_0-9]*|SPI_[A-Z_0-9a-z]+)\b" | sort -u | tr '\n' ' '
