"""placeholder replaced below"""
